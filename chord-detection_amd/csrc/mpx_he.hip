// Harmonic-Energy chroma (reference method 2) as one fused HIP kernel per frame:
//   frame cut (index math only, never materialised)      dsp/frame.py:5-14
//   x * hamming_sym(N) -> rfft -> sqrt(|X|)               harmonic_energy.py:42-43
//   12 x octave x harmonic half-open bin-window maxima,
//   weighted 1/h, summed per pitch class                  harmonic_energy.py:44-67
// plus the cross-frame Chromagram accumulation            chromagram.py:42-45
//
// Data layout in HBM: the signal is one contiguous fp32 array; frame f reads
// samples [f*hop, f*hop+N) (or the range given by its FrameDesc) straight from
// it, so overlapped frames re-use each other's cache lines in the XCD-local L2
// (the blockIdx -> frame map hands every XCD a contiguous run of frames).
// Output: [F,12] doubles.  Nothing else touches HBM: the N-point real FFT is an
// N/2-point complex FFT held entirely in LDS (mpx_fft.hpp) followed by the
// real-split butterfly evaluated only for the bins the 48 windows look at.
#include <cmath>
#include <cstdarg>

#include "mpx_fft.hpp"
#include "mpx_internal.hpp"

namespace mpx {

template <typename Real>
struct HeArgs {
    const float* sig;
    long long n;            // samples in sig (hop mode)
    const FrameDesc* desc;  // nullptr => frame f starts at f*hop
    long long num_frames;
    int hop;
    const Real* window;     // [N]
    const cx<Real>* tw;     // [M]
    const cx<Real>* twn;    // [M+1]
    const int* wk0;
    const int* wk1;
    const Real* ww;
    int nwin, wins_per_note, num_harmonic;
    int kmin, kmax;
    double* out;            // [F,12]
};

// XCD-aware bijective remap: workgroup b runs on XCD b%8 (observed dispatch
// order; a wrong guess only costs speed).  Give each XCD a contiguous block of
// frames so the (N-hop)-sample overlap between neighbours hits its own L2.
__device__ __forceinline__ long long xcd_contiguous(long long b, long long g) {
    const long long q = g >> 3, r = g & 7;
    const long long xcd = b & 7, slot = b >> 3;
    const long long base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

template <int N, int T, typename Real>
__global__ __launch_bounds__(T) void he_kernel(HeArgs<Real> a) {
    constexpr int M = N / 2;
    constexpr int EPT = M / T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cx<Real>* buf = reinterpret_cast<cx<Real>*>(smem);
    Real* mag = reinterpret_cast<Real*>(smem + sizeof(cx<Real>) * lds_slots(M));
    const int nmag = a.kmax - a.kmin;
    Real* winmax = mag + nmag;

    const int tid = threadIdx.x;
    const long long f = xcd_contiguous(blockIdx.x, gridDim.x);
    long long start;
    int valid;
    if (a.desc) {
        start = a.desc[f].start;
        valid = a.desc[f].valid;
    } else {
        start = f * (long long)a.hop;
        const long long left = a.n - start;
        valid = left >= N ? N : (left > 0 ? (int)left : 0);
    }
    const float* __restrict__ x = a.sig + start;

    // windowed load, two real samples packed per complex point
    cx<Real> regs[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int p = first_pass_index<M, T>(tid, e);
        const int s = 2 * p;
        const Real x0 = s < valid ? (Real)x[s] : (Real)0;
        const Real x1 = s + 1 < valid ? (Real)x[s + 1] : (Real)0;
        regs[e] = {x0 * a.window[s], x1 * a.window[s + 1]};
    }
    fft_lds<M, T, true, Real>(buf, a.tw, regs, tid);

    // real-split: X[k] = E + (-i) W_N^k D, E=(Z[k]+conj Z[M-k])/2, D=(Z[k]-conj Z[M-k])/2
    for (int i = tid; i < nmag; i += T) {
        const int k = a.kmin + i;
        const cx<Real> A = buf[lds_slot(k & (M - 1))];
        cx<Real> B = buf[lds_slot((M - k) & (M - 1))];
        B.y = -B.y;
        const cx<Real> E = {(Real)0.5 * (A.x + B.x), (Real)0.5 * (A.y + B.y)};
        const cx<Real> D = {(Real)0.5 * (A.x - B.x), (Real)0.5 * (A.y - B.y)};
        const cx<Real> X = cadd(E, mul_mi(cmul(a.twn[k], D)));
        mag[i] = sqrt(sqrt(X.x * X.x + X.y * X.y));
    }
    __syncthreads();
    // half-open window maxima (harmonic_energy.py:58-62)
    for (int w = tid; w < a.nwin; w += T) {
        Real m = -INFINITY;
        for (int k = a.wk0[w]; k < a.wk1[w]; ++k) {
            const Real v = mag[k - a.kmin];
            m = v > m ? v : m;
        }
        winmax[w] = m;
    }
    __syncthreads();
    // chroma[n] = sum_octave ( sum_harmonic max/h ), same association as the reference
    if (tid < 12) {
        double chroma = 0.0;
        const int base = tid * a.wins_per_note;
        for (int o = 0; o < a.wins_per_note; o += a.num_harmonic) {
            double note_sum = 0.0;
            for (int h = 0; h < a.num_harmonic; ++h)
                note_sum += (double)winmax[base + o + h] * (double)a.ww[base + o + h];
            chroma += note_sum;
        }
        a.out[f * 12 + tid] = chroma;
    }
}

// ------------------------------------------------------------------ reductions
// Single segment, stage 1: chunk c sums frames [c*CH, (c+1)*CH) -> partial[c][12]
constexpr int SUM_CH = 256;
__global__ __launch_bounds__(64) void sum_chunks_kernel(const double* __restrict__ frames,
                                                        long long num_frames, double* partial) {
    __shared__ double sh[5][12];
    const int lane = threadIdx.x;
    const long long f0 = (long long)blockIdx.x * SUM_CH;
    long long f1 = f0 + SUM_CH;
    if (f1 > num_frames) f1 = num_frames;
    if (lane < 60) {
        const int bin = lane % 12, sub = lane / 12;
        double acc = 0.0;
        for (long long f = f0 + sub; f < f1; f += 5) acc += frames[f * 12 + bin];
        sh[sub][bin] = acc;
    }
    __syncthreads();
    if (lane < 12)
        partial[(long long)blockIdx.x * 12 + lane] =
            (((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane]) + sh[4][lane];
}

// One workgroup per segment: out[s] = sum of rows [seg[s], seg[s+1]) of `rows`.
__global__ __launch_bounds__(64) void sum_segments_kernel(const double* __restrict__ rows,
                                                          const long long* __restrict__ seg,
                                                          double* out) {
    __shared__ double sh[5][12];
    const int lane = threadIdx.x;
    const long long f0 = seg[blockIdx.x], f1 = seg[blockIdx.x + 1];
    if (lane < 60) {
        const int bin = lane % 12, sub = lane / 12;
        double acc = 0.0;
        for (long long f = f0 + sub; f < f1; f += 5) acc += rows[f * 12 + bin];
        sh[sub][bin] = acc;
    }
    __syncthreads();
    if (lane < 12)
        out[(long long)blockIdx.x * 12 + lane] =
            (((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane]) + sh[4][lane];
}

// Same, single segment [0, n) without a segment table.
__global__ __launch_bounds__(64) void sum_all_kernel(const double* __restrict__ rows, long long n,
                                                     double* out) {
    __shared__ double sh[5][12];
    const int lane = threadIdx.x;
    if (lane < 60) {
        const int bin = lane % 12, sub = lane / 12;
        double acc = 0.0;
        for (long long f = sub; f < n; f += 5) acc += rows[f * 12 + bin];
        sh[sub][bin] = acc;
    }
    __syncthreads();
    if (lane < 12)
        out[lane] = (((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane]) + sh[4][lane];
}

int segment_sum(mpx_ctx* ctx, const double* d_frames, const long long* d_seg, int num_seg,
                int64_t num_frames, double* d_out, hipStream_t stream) {
    if (d_seg) {
        if (num_seg > 0)
            hipLaunchKernelGGL(sum_segments_kernel, dim3(num_seg), dim3(64), 0, stream, d_frames, d_seg,
                               d_out);
    } else if (num_frames <= 2 * SUM_CH) {
        hipLaunchKernelGGL(sum_all_kernel, dim3(1), dim3(64), 0, stream, d_frames, (long long)num_frames,
                           d_out);
    } else {
        const long long nch = (num_frames + SUM_CH - 1) / SUM_CH;
        int rc = ensure(ctx, ctx->d_partials, (size_t)nch * 12 * sizeof(double));
        if (rc) return rc;
        double* part = (double*)ctx->d_partials.p;
        hipLaunchKernelGGL(sum_chunks_kernel, dim3((unsigned)nch), dim3(64), 0, stream, d_frames,
                           (long long)num_frames, part);
        hipLaunchKernelGGL(sum_all_kernel, dim3(1), dim3(64), 0, stream, part, nch, d_out);
    }
    MPX_HIP(ctx, hipGetLastError());
    return MPX_OK;
}

// ------------------------------------------------------------------ plan build
static double he_round_half_even(double v) { return std::nearbyint(v); }

template <typename Real>
static int he_build_plan(mpx_ctx* ctx, int fs, int N, const mpx_he_params& p, HePlan& plan) {
    const int M = N / 2;
    // note table: librosa.cqt_frequencies(12, fmin=note_to_hz('C3')) harmonic_energy.py:33
    const double c3 = 440.0 * std::pow(2.0, (48.0 - 69.0) / 12.0);
    const double divisor_ratio = (fs / 4.0) / N;  // harmonic_energy.py:35 (quirk A.2)
    std::vector<int> k0, k1;
    std::vector<Real> ww;
    int kmin = 1 << 30, kmax = -(1 << 30);
    for (int n = 0; n < 12; ++n) {
        const double note = c3 * std::pow(2.0, n / 12.0);
        for (int oct = 1; oct <= p.num_octave; ++oct)
            for (int h = 1; h <= p.num_harmonic; ++h) {
                const double kp = he_round_half_even((note * oct * h) / divisor_ratio);
                const int a = (int)(kp - p.num_bins * h), b = (int)(kp + p.num_bins * h);
                k0.push_back(a);
                k1.push_back(b);
                ww.push_back((Real)(1.0 / h));
                if (b > a) {
                    kmin = a < kmin ? a : kmin;
                    kmax = b > kmax ? b : kmax;
                }
            }
    }
    if (kmin > kmax) kmin = kmax = 0;
    if (kmin < 0 || kmax > M + 1)
        return set_error(ctx, MPX_EINVAL,
                         "harmonic-energy window [%d,%d) outside the %d-bin spectrum (the reference "
                         "raises IndexError / wraps here)", kmin, kmax, M + 1);
    plan.nwin = (int)k0.size();
    plan.wins_per_note = p.num_octave * p.num_harmonic;
    plan.num_harmonic = p.num_harmonic;
    plan.kmin = kmin;
    plan.kmax = kmax;
    std::vector<Real> win(N);
    for (int i = 0; i < N; ++i)  // scipy.signal.hamming(N), symmetric
        win[i] = (Real)(0.54 - 0.46 * std::cos(2.0 * M_PI * i / (double)(N - 1)));
    std::vector<cx<Real>> tw(M), twn(M + 1);
    for (int j = 0; j < M; ++j) {
        const long double ang = -2.0L * M_PIl * j / (long double)M;
        tw[j] = {(Real)cosl(ang), (Real)sinl(ang)};
    }
    for (int k = 0; k <= M; ++k) {
        const long double ang = -2.0L * M_PIl * k / (long double)N;
        twn[k] = {(Real)cosl(ang), (Real)sinl(ang)};
    }
    plan.window = upload(ctx, win.data(), win.size() * sizeof(Real));
    plan.tw = upload(ctx, tw.data(), tw.size() * sizeof(cx<Real>));
    plan.twn = upload(ctx, twn.data(), twn.size() * sizeof(cx<Real>));
    plan.wk0 = (int*)upload(ctx, k0.data(), k0.size() * sizeof(int));
    plan.wk1 = (int*)upload(ctx, k1.data(), k1.size() * sizeof(int));
    plan.ww = upload(ctx, ww.data(), ww.size() * sizeof(Real));
    if (!plan.window || !plan.tw || !plan.twn || !plan.wk0 || !plan.wk1 || !plan.ww) return MPX_ENOMEM;
    return MPX_OK;
}

template <int N, int T, typename Real>
static int he_launch(mpx_ctx* ctx, const HePlan& plan, const float* d_signal, int64_t n,
                     const FrameDesc* d_desc, int64_t num_frames, int hop, double* d_out,
                     hipStream_t stream) {
    HeArgs<Real> a;
    a.sig = d_signal;
    a.n = n;
    a.desc = d_desc;
    a.num_frames = num_frames;
    a.hop = hop;
    a.window = (const Real*)plan.window;
    a.tw = (const cx<Real>*)plan.tw;
    a.twn = (const cx<Real>*)plan.twn;
    a.wk0 = plan.wk0;
    a.wk1 = plan.wk1;
    a.ww = (const Real*)plan.ww;
    a.nwin = plan.nwin;
    a.wins_per_note = plan.wins_per_note;
    a.num_harmonic = plan.num_harmonic;
    a.kmin = plan.kmin;
    a.kmax = plan.kmax;
    a.out = d_out;
    const size_t lds = sizeof(cx<Real>) * lds_slots(N / 2) + sizeof(Real) * (size_t)(plan.kmax - plan.kmin + plan.nwin);
    if (lds > 160 * 1024)
        return set_error(ctx, MPX_EUNSUPPORTED, "frame %d needs %zu B of LDS (> 160 KiB)", N, lds);
    auto kern = he_kernel<N, T, Real>;
    if (lds > 64 * 1024)
        MPX_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)num_frames), dim3(T), lds, stream, a);
    MPX_HIP(ctx, hipGetLastError());
    return MPX_OK;
}

template <typename Real>
static int he_dispatch(mpx_ctx* ctx, const HePlan& plan, const float* d_signal, int64_t n,
                       const FrameDesc* d_desc, int64_t num_frames, int frame, int hop, double* d_out,
                       hipStream_t stream) {
    switch (frame) {
        case 1024: return he_launch<1024, 64, Real>(ctx, plan, d_signal, n, d_desc, num_frames, hop, d_out, stream);
        case 2048: return he_launch<2048, 64, Real>(ctx, plan, d_signal, n, d_desc, num_frames, hop, d_out, stream);
        case 4096: return he_launch<4096, 128, Real>(ctx, plan, d_signal, n, d_desc, num_frames, hop, d_out, stream);
        case 8192: return he_launch<8192, 256, Real>(ctx, plan, d_signal, n, d_desc, num_frames, hop, d_out, stream);
        case 16384: return he_launch<16384, 512, Real>(ctx, plan, d_signal, n, d_desc, num_frames, hop, d_out, stream);
        default:
            return set_error(ctx, MPX_EUNSUPPORTED,
                             "harmonic energy: frame size %d is not a power of two in [1024, 16384]", frame);
    }
}

int he_run(mpx_ctx* ctx, const float* d_signal, int64_t n, const FrameDesc* d_desc, int64_t num_frames,
           int fs, const mpx_he_params* params, int frame, int hop, double* d_chroma_frames,
           hipStream_t stream) {
    mpx_he_params p = params ? *params : mpx_he_params{2, 2, 2};
    if (p.num_harmonic < 1 || p.num_octave < 1 || p.num_bins < 0 || p.num_harmonic * p.num_octave > 64)
        return set_error(ctx, MPX_EINVAL, "bad harmonic-energy params (%d,%d,%d)", p.num_harmonic,
                         p.num_octave, p.num_bins);
    if (fs <= 0) return set_error(ctx, MPX_EINVAL, "fs must be positive");
    if (frame < 1024 || frame > 16384 || (frame & (frame - 1)))
        return set_error(ctx, MPX_EUNSUPPORTED,
                         "harmonic energy: frame size %d is not a power of two in [1024, 16384]", frame);
    if (num_frames == 0) return MPX_OK;
    const bool f32 = ctx->flags & MPX_FLAG_F32;
    auto key = std::make_tuple(fs, frame, p.num_harmonic, p.num_octave, p.num_bins);
    auto it = ctx->he_plans.find(key);
    if (it == ctx->he_plans.end()) {
        HePlan plan;
        int rc = f32 ? he_build_plan<float>(ctx, fs, frame, p, plan) : he_build_plan<double>(ctx, fs, frame, p, plan);
        if (rc) return rc;
        it = ctx->he_plans.emplace(key, plan).first;
    }
    return f32 ? he_dispatch<float>(ctx, it->second, d_signal, n, d_desc, num_frames, frame, hop, d_chroma_frames, stream)
               : he_dispatch<double>(ctx, it->second, d_signal, n, d_desc, num_frames, frame, hop, d_chroma_frames, stream);
}

}  // namespace mpx
