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
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <string>
#include <cstring>
#include <cstdarg>
#include <type_traits>

#include "mpx_fft.hpp"
#include "mpx_fft_dif.hpp"
#include "mpx_internal.hpp"
#include "mpx_he_wave.hpp"

namespace mpx {

template <typename Real>
struct HeArgs {
    const float* sig;
    long long n;            // samples in sig (hop mode)
    const FrameDesc* desc;  // nullptr => frame f starts at f*hop
    long long num_frames;
    int hop;
    const cx<Real>* wbase;  // [2T]  (cos, sin)(2*pi*s/(N-1)), s < 2T: this thread's two base samples
    const cx<Real>* woffs;  // [EPT] (cos, sin)(2*pi*o_e/(N-1)), o_e = 2*e*T: staged in LDS by the kernel
    const cx<Real>* tw;     // [M]
    const cx<Real>* twn;    // [M+1]
    const int* wk0;   // windows as index ranges into bins[]
    const int* wk1;
    const unsigned* slots;  // per window bin k: LDS slot of Z[k] | slot of Z[M-k] << 16
    const cx<Real>* twnb;  // W_N^k at bins[i], indexed like bins
    int nb;
    const Real* ww;
    int nwin, wins_per_note, num_harmonic;
    int kmin, kmax;
    double* out;            // [F,12] per-frame chroma, may be null
    double* partial;        // [gridDim.x,12] per-workgroup sums over its frames, may be null
    double* sum;            // [12] total over all frames (written by the last workgroup to finish)
    unsigned* counter;      // arrival ticket for that hand-off; zero before and after every launch
};


// This thread's EPT sample pairs of one frame (pair e starts at sample 2*(tid + e*STRIDE)); zero
// beyond `valid` (the frame_cutter padding).  Full, 8-byte aligned frames -- all but the tail of
// a signal -- take the branch-free path: one base address plus immediate offsets.
template <int EPT, int STRIDE, bool FAST>
__device__ __forceinline__ void load_frame(float2* raw, const float* __restrict__ x, int tid, int valid) {
    const float* p = x + 2 * tid;
    if (FAST) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) raw[e] = *reinterpret_cast<const float2*>(p + 2 * e * STRIDE);
    } else {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int s = 2 * (tid + e * STRIDE);
            raw[e].x = s < valid ? p[2 * e * STRIDE] : 0.f;
            raw[e].y = s + 1 < valid ? p[2 * e * STRIDE + 1] : 0.f;
        }
    }
}

// Persistent workgroups: workgroup w owns the contiguous frame range
// [w*per, (w+1)*per).  Everything that does not depend on the frame (this
// thread's window samples, its twiddle bases) is loaded once and stays in
// registers; the next frame's samples are prefetched while the current frame is
// in the FFT; the 12-bin chroma is accumulated across the workgroup's frames so
// that only one [12] partial per workgroup (plus the optional per-frame rows)
// goes back to HBM.
// Which FFT engine a (N, T) instance uses: the in-place wave-local DIF (8 points per thread, M <= 4096)
// or the generic Stockham engine.
template <int N, int T>
constexpr bool he_uses_dif() { return (N / 2) <= 4096 && (N / 2) / T == 8; }
template <int N, int T>
constexpr int he_buf_slots() { return he_uses_dif<N, T>() ? N / 2 : lds_slots(N / 2); }

// Waves per SIMD the LDS footprint allows (the register allocator is told to stay inside it).
template <int N, int T, typename Real>
constexpr int he_waves_per_simd() {
    constexpr int lds = (int)sizeof(cx<Real>) * he_buf_slots<N, T>() + 4096 + 256;
    constexpr int blocks = (160 * 1024 / lds) > 8 ? 8 : (160 * 1024 / lds);
    constexpr int w = blocks * T / 256;
    constexpr int cap = (N / 2 / T) > 8 ? 2 : 4;  // 16 points per thread need the registers of <= 2 waves/SIMD
    return w < 1 ? 1 : (w > cap ? cap : w);
}

template <int N, int T, typename Real>
__global__ __launch_bounds__(T, (he_waves_per_simd<N, T, Real>())) void he_kernel(HeArgs<Real> a) {
    constexpr int M = N / 2;
    constexpr int EPT = M / T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool DIF = he_uses_dif<N, T>();
    cx<Real>* buf = reinterpret_cast<cx<Real>*>(smem);
    // behind the FFT buffer, at compile-time offsets: [window offset angles: EPT complex][flag: 16 B], then mag / winmax
    cx<Real>* woffs_lds = buf + he_buf_slots<N, T>();
    char* tail = reinterpret_cast<char*>(woffs_lds + EPT);
    Real* mag = reinterpret_cast<Real*>(tail + 16);
    const int nmag = a.nb;  // magnitudes are computed for the bins the windows touch only (compact, window order)
    Real* winmax = mag + nmag;
    // the window table (index ranges into mag[], 1/harmonic weights) behind winmax: wave 0 walks it every frame, and
    // from global memory each dependent step of that walk costs an L2 round trip
    Real* ww_lds = winmax + a.nwin;
    int* wk_lds = reinterpret_cast<int*>(ww_lds + a.nwin);
    unsigned* slots_lds = reinterpret_cast<unsigned*>(wk_lds + 2 * a.nwin);  // [nb] see HeArgs::slots

    const int tid = threadIdx.x;
    if (tid < EPT) woffs_lds[tid] = a.woffs[tid];
    for (int i = tid; i < a.nwin; i += T) {
        ww_lds[i] = a.ww[i];
        wk_lds[2 * i] = a.wk0[i];
        wk_lds[2 * i + 1] = a.wk1[i];
    }
    for (int i = tid; i < a.nb; i += T) slots_lds[i] = a.slots[i];
    __syncthreads();
    const long long w = xcd_contiguous(blockIdx.x, gridDim.x);
    const long long per = (a.num_frames + gridDim.x - 1) / gridDim.x;
    const long long f0 = w * per;
    long long f1 = f0 + per;
    if (f1 > a.num_frames) f1 = a.num_frames;

    // Loop invariants.  The symmetric Hamming window 0.54 - 0.46 cos(2 pi s/(N-1)) at this thread's
    // samples s = 2*tid + j + o_e is rebuilt per frame from the angle-addition formula: 4 doubles per
    // thread (cos/sin at 2*tid, 2*tid+1) + EPT wave-uniform pairs, instead of 2*EPT doubles per thread.
    const cx<Real> wb0 = a.wbase[2 * tid], wb1 = a.wbase[2 * tid + 1];
    // first-pass read order: register e holds complex point tid + e*FSTRIDE (one butterfly per thread)
    constexpr int FSTRIDE = M / EPT;
    static_assert(Plan<M, EPT>::radix(0) == EPT && FSTRIDE == T, "first pass must be one radix-EPT butterfly per thread");

    auto frame_span = [&](long long f, long long& start, int& valid) {
        if (a.desc) {
            start = a.desc[f].start;
            valid = a.desc[f].valid;
        } else {
            start = f * (long long)a.hop;
            const long long left = a.n - start;
            valid = left >= N ? N : (left > 0 ? (int)left : 0);
        }
    };

    // twiddle bases of the DIF passes: loop invariant, 1 complex per pass
    DifTwiddles<DIF ? M : 512, Real> twd;
    if constexpr (DIF) twd = dif_load_twiddles<M, Real>(a.tw, tid);
    double acc = 0.0;  // chroma bin `wl` summed over this workgroup's frames, in lane wl < 12 of the LAST wave

    // Every frame of this workgroup full-length and 8-byte aligned (true for all but the workgroup holding
    // the ragged tail of a signal, and for packed clips of odd length)?  Then the frame loop is instantiated
    // without any per-sample predicate: the prefetch is 8 plain loads that stay in flight across the FFT.
    bool fast = f0 < f1 && (reinterpret_cast<uintptr_t>(a.sig) & 7) == 0;
    if (fast) {
        if (a.desc) {
            for (long long f = f0; f < f1; ++f) fast = fast && a.desc[f].valid >= N && (a.desc[f].start & 1) == 0;
        } else {
            fast = (a.hop & 1) == 0 && (f1 - 1) * (long long)a.hop + N <= a.n;
        }
    }
    // Which of this thread's EPT last-pass outputs does anyone look at?  Bins [kmin,kmax), their mirrors
    // M-k and bin 0.  Thread-invariant: one bit mask, computed once.
    unsigned need = 0;
    {
        const int lo2 = M - a.kmax + 1, hi2 = M - a.kmin;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            int q;
            if constexpr (DIF) {
                constexpr int RL = DifPlan<M>::radix(DifPlan<M>::n - 1);
                q = dif_freq<M>(dif_last_pos<M>(tid, e / RL, e % RL));
            } else {
                q = last_pass_index<M, T>(tid, e);
            }
            if ((q >= a.kmin && q < a.kmax) || (q >= lo2 && q <= hi2) || q == 0) need |= 1u << e;
        }
    }
    auto run = [&](auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    float2 raw[EPT];
    if (f0 < f1) {
        long long start;
        int valid;
        frame_span(f0, start, valid);
        load_frame<EPT, FSTRIDE, FAST>(raw, a.sig + start, tid, valid);
    }
    for (long long f = f0; f < f1; ++f) {
        // opaque copy of the thread id: LDS addresses / store predicates derived from it are rebuilt every
        // frame (a few integer ops) instead of being hoisted into dozens of live registers
        int ot = tid;
        asm volatile("" : "+v"(ot));
        cx<Real> regs[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const cx<Real> o = woffs_lds[e];  // same address in every lane: one broadcast LDS read
            // 0.54 - 0.46 cos(a + b), with -0.46 folded into the offset table: two FMAs per sample
            const Real w0 = wb0.x * o.x + ((Real)0.54 - wb0.y * o.y);
            const Real w1 = wb1.x * o.x + ((Real)0.54 - wb1.y * o.y);
            regs[e] = {(Real)raw[e].x * w0, (Real)raw[e].y * w1};
        }
        {
            // Prefetch the next frame; its latency hides under this frame's FFT.  Unconditional on purpose
            // (the last iteration re-reads its own frame): a conditional prefetch makes the compiler drain
            // vmcnt(0) at the join right behind the loads, which exposes their full latency every frame.
            long long start;
            int valid;
            frame_span(f + 1 < f1 ? f + 1 : f, start, valid);
            load_frame<EPT, FSTRIDE, FAST>(raw, a.sig + start, tid, valid);
        }
        // FFT; the last pass keeps its outputs in registers and only the bins the windows look at are stored
        if constexpr (DIF) {
            {
                // dif_fft_keep_last, spelled out so that the first pass and the prefetch use the real thread id (their
                // two address registers stay live across frames) while the later passes rebuild theirs from `ot`:
                // hoisting those as well overflows the 128-register budget of 4 waves/SIMD into scratch
                using PL = DifPlan<M>;
                dif_butterfly<M, 0, false, true, Real>(buf, twd, regs, tid);
                __syncthreads();
                dif_middle<M, 1, Real>(buf, twd, regs, ot);
                constexpr int RLAST = PL::radix(PL::n - 1);
#pragma unroll
                for (int h = 0; h < 8 / RLAST; ++h)
                    dif_butterfly<M, PL::n - 1, true, false, Real>(buf, twd, regs + h * RLAST, dif_bid<M, PL::n - 1>(ot, h));
            }
            constexpr int RL = DifPlan<M>::radix(DifPlan<M>::n - 1);
#pragma unroll
            for (int e = 0; e < EPT; ++e)
                if (need & (1u << e)) buf[sigma<M>(dif_last_pos<M>(ot, e / RL, e % RL))] = regs[e];
        } else {
            fft_lds_keep_last<M, T, true, Real>(buf, a.tw, regs, tid);
#pragma unroll
            for (int e = 0; e < EPT; ++e)
                if (need & (1u << e)) buf[lds_slot(last_pass_index<M, T>(tid, e))] = regs[e];
        }
        // W_N^k of this thread's first window bin: the load is issued before the barrier, its L2 latency passes
        // while the workgroup gathers (entry 0 always exists)
        const cx<Real> tw_first = a.twnb[ot < nmag ? ot : 0];
        __syncthreads();

        // real-split: X[k] = E + (-i) W_N^k D, E=(Z[k]+conj Z[M-k])/2, D=(Z[k]-conj Z[M-k])/2
        // (LDS slots of Z[k] and Z[M-k] come from a table built with the plan: no index arithmetic here.)  mag[]
        // holds |X|^2: the fourth root is monotone, so it is taken of the 48 window maxima only, not of every bin.
        auto split_bin = [&](int i, cx<Real> twk) {
            const unsigned s = slots_lds[i];
            const cx<Real> A = buf[s & 0xffffu];
            cx<Real> B = buf[s >> 16];
            B.y = -B.y;
            const cx<Real> E = {(Real)0.5 * (A.x + B.x), (Real)0.5 * (A.y + B.y)};
            const cx<Real> D = {(Real)0.5 * (A.x - B.x), (Real)0.5 * (A.y - B.y)};
            const cx<Real> X = cadd(E, mul_mi(cmul(twk, D)));
            mag[i] = X.x * X.x + X.y * X.y;
        };
        if (ot < nmag) split_bin(ot, tw_first);
        for (int i = ot + T; i < nmag; i += T) split_bin(i, a.twnb[i]);
        __syncthreads();  // mag[] complete, and nobody reads buf any more: the next frame may overwrite it
        // Window maxima and pitch-class sums by the LAST wave alone (the first ones also take the second round of
        // the split loop above when nb > T): its LDS traffic is ordered by the wave's own in-order LDS pipeline, so
        // no further workgroup barrier is needed and the other waves run ahead into the next frame (they meet this
        // wave again at that frame's first barrier, before mag[] is rewritten).  Everything on this path is a
        // serial chain on the workgroup's critical path, hence tables in LDS and independent loads.
        const int wl = ot - (T - 64);  // lane of the last wave (rebuilt from the opaque id: one register less across the FFT)
        if (wl >= 0) {
            for (int wi = wl; wi < a.nwin; wi += 64) {  // half-open window maxima (harmonic_energy.py:58-62)
                Real m = -INFINITY;
                const int k1 = wk_lds[2 * wi + 1], kl = k1 - 1;
                for (int k = wk_lds[2 * wi]; k < k1; k += 4) {
                    // four independent LDS reads per round (indices clamped to the window's last bin: seeing a bin
                    // twice does not change a maximum), compared in bin order like the reference's loop
                    const Real v0 = mag[k], v1 = mag[k + 1 < kl ? k + 1 : kl], v2 = mag[k + 2 < kl ? k + 2 : kl],
                               v3 = mag[k + 3 < kl ? k + 3 : kl];
                    m = v0 > m ? v0 : m;
                    m = v1 > m ? v1 : m;
                    m = v2 > m ? v2 : m;
                    m = v3 > m ? v3 : m;
                }
                // sqrt(|X|) = (|X|^2)^(1/4) of the maximum (harmonic_energy.py:43); an empty window keeps -inf
                winmax[wi] = m < (Real)0 ? m : sqrt(sqrt(m));
            }
            wave_lds_fence();
            // chroma[n] = sum_octave ( sum_harmonic max/h ), same association as the reference
            if (wl < 12) {
                double chroma = 0.0;
                const int base = wl * a.wins_per_note;
                if (a.wins_per_note == 4 && a.num_harmonic == 2) {  // the reference's defaults: all eight reads at once
                    const double m0 = winmax[base], m1 = winmax[base + 1], m2 = winmax[base + 2], m3 = winmax[base + 3];
                    const double w0 = ww_lds[base], w1 = ww_lds[base + 1], w2 = ww_lds[base + 2], w3 = ww_lds[base + 3];
                    double n0 = 0.0, n1 = 0.0;
                    n0 += m0 * w0;
                    n0 += m1 * w1;
                    n1 += m2 * w2;
                    n1 += m3 * w3;
                    chroma += n0;
                    chroma += n1;
                } else {
                    for (int o = 0; o < a.wins_per_note; o += a.num_harmonic) {
                        double note_sum = 0.0;
                        for (int h = 0; h < a.num_harmonic; ++h)
                            note_sum += (double)winmax[base + o + h] * (double)ww_lds[base + o + h];
                        chroma += note_sum;
                    }
                }
                if (a.out) a.out[f * 12 + wl] = chroma;
                acc += chroma;
            }
            wave_lds_fence();
        }
        // the next iteration's first LDS write (pass 1) is ordered after this iteration's last
        // LDS read of buf by the two barriers above; winmax/mag are rewritten only after the
        // FFT's own barriers
    }
    };  // run
    if (fast)
        run(std::true_type{});
    else
        run(std::false_type{});
    if (a.partial) {
        // Cross-workgroup reduction inside the launch, two levels of "last arriver adds": every workgroup
        // publishes its 12 partial sums; the last of each group of 32 workgroups adds that group's rows and
        // publishes a group row; the last group to finish adds the group rows.  Fixed summation order ->
        // deterministic; two short dependent reads instead of one long one at the tail of the kernel.
        // Protocol per hop (cdna guide G16, form R1): write-through (sc1) 8-byte stores -> vmcnt(0) in the
        // storing wave -> barrier -> one lane takes a relaxed agent-scope ticket; readers use sc1
        // (L1-bypassing) loads, so no release/acquire fence is needed.  Counters are left at zero.
        int* flag = reinterpret_cast<int*>(tail);
        using u64 = unsigned long long;
        constexpr int GROUP = 32;
        u64* part = reinterpret_cast<u64*>(a.partial);
        const long long g = gridDim.x;
        const long long ngroups = (g + GROUP - 1) / GROUP;
        const long long grp = w / GROUP;
        const long long grp_first = grp * GROUP;
        const long long grp_size = (g - grp_first) < GROUP ? (g - grp_first) : GROUP;
        double* sh = reinterpret_cast<double*>(smem);  // buf is dead
        if (T > 64) {  // the publishing lanes below are tid < 12: hand them the sums of the last wave's lanes
            if (tid >= T - 64 && tid < T - 52) sh[tid - (T - 64)] = acc;
            __syncthreads();
            if (tid < 12) acc = sh[tid];
            __syncthreads();
        }

        // sum `count` rows starting at `first`: 12 bins x up to 21 strided sub-sums, all loads in flight at once
        auto sum_rows = [&](long long first, long long count) -> double {
            constexpr int SUBS = T / 12 < 21 ? T / 12 : 21;
            if (tid < 12 * SUBS) {
                const int bin = tid % 12, sub = tid / 12;
                double s0 = 0.0, s1 = 0.0;
                long long r = sub;
                for (; r + SUBS < count; r += 2 * SUBS) {
                    const u64 v0 = __hip_atomic_load(part + (first + r) * 12 + bin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const u64 v1 = __hip_atomic_load(part + (first + r + SUBS) * 12 + bin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s0 += __longlong_as_double((long long)v0);
                    s1 += __longlong_as_double((long long)v1);
                }
                if (r < count)
                    s0 += __longlong_as_double((long long)__hip_atomic_load(part + (first + r) * 12 + bin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                sh[sub * 12 + bin] = s0 + s1;
            }
            __syncthreads();
            double t = 0.0;
            if (tid < 12)
                for (int s2 = 0; s2 < SUBS; ++s2) t += sh[s2 * 12 + tid];
            __syncthreads();
            return t;
        };
        auto publish_and_ticket = [&](long long row, double value, unsigned* counter, unsigned last_ticket) -> bool {
            if (tid < 12) __hip_atomic_store(part + row * 12 + tid, (u64)__double_as_longlong(value), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int last = (t == last_ticket);
                if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *flag = last;
            }
            __syncthreads();
            const bool last = *flag != 0;
            __syncthreads();
            return last;
        };
        if (publish_and_ticket(w, acc, a.counter + 1 + grp, (unsigned)(grp_size - 1))) {
            const double gsum = sum_rows(grp_first, grp_size);
            if (publish_and_ticket(g + grp, gsum, a.counter, (unsigned)(ngroups - 1))) {
                const double total = sum_rows(g, ngroups);
                if (tid < 12) a.sum[tid] = total;
            }
        }
    }
}

// ------------------------------------------------------------------ reductions
// Single segment, stage 1: chunk c sums frames [c*CH, (c+1)*CH) -> partial[c][12]
constexpr int SUM_CH = 256;
#ifndef HE4096_T
#define HE4096_T 256
#endif
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

// Same, single segment [0, n) without a segment table: 252 lanes = 12 bins x 21 strided
// sub-sums, combined in a fixed order (deterministic).
__global__ __launch_bounds__(256) void sum_all_kernel(const double* __restrict__ rows, long long n,
                                                      double* out) {
    __shared__ double sh[21][12];
    const int lane = threadIdx.x;
    if (lane < 252) {
        const int bin = lane % 12, sub = lane / 12;
        double acc = 0.0;
        // eight rows requested at once, added in row order (the same additions as one at a time: a launch over the 256
        // workgroup sums of he_wave_kernel is 13 dependent L2 round trips otherwise)
        for (long long f = sub; f < n; f += 21 * 8) {
            double v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = f + 21 * j < n ? rows[(f + 21 * j) * 12 + bin] : 0.0;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (f + 21 * j < n) acc += v[j];
        }
        sh[sub][bin] = acc;
    }
    __syncthreads();
    if (lane < 12) {
        double t = 0.0;
#pragma unroll
        for (int s2 = 0; s2 < 21; ++s2) t += sh[s2][lane];
        out[lane] = t;
    }
}

int segment_sum(mpx_ctx* ctx, const double* d_frames, const long long* d_seg, int num_seg,
                int64_t num_frames, double* d_out, hipStream_t stream) {
    if (d_seg) {
        if (num_seg > 0)
            hipLaunchKernelGGL(sum_segments_kernel, dim3(num_seg), dim3(64), 0, stream, d_frames, d_seg,
                               d_out);
    } else if (num_frames <= 2 * SUM_CH) {
        hipLaunchKernelGGL(sum_all_kernel, dim3(1), dim3(256), 0, stream, d_frames, (long long)num_frames,
                           d_out);
    } else {
        const long long nch = (num_frames + SUM_CH - 1) / SUM_CH;
        int rc = ensure(ctx, ctx->d_partials, (size_t)nch * 12 * sizeof(double));
        if (rc) return rc;
        double* part = (double*)ctx->d_partials.p;
        hipLaunchKernelGGL(sum_chunks_kernel, dim3((unsigned)nch), dim3(64), 0, stream, d_frames,
                           (long long)num_frames, part);
        hipLaunchKernelGGL(sum_all_kernel, dim3(1), dim3(256), 0, stream, part, nch, d_out);
    }
    MPX_HIP(ctx, hipGetLastError());
    return MPX_OK;
}

// ------------------------------------------------------------------ plan build
static double he_round_half_even(double v) { return std::nearbyint(v); }

// The windows of harmonic_energy.py:44-62 on an nbins-bin spectrum: bin numbers [k0, k1) per (note, octave, harmonic), the
// weights 1/h, and the COMPACT BIN LIST the kernels evaluate -- every window is a run [c0, c1) of it.  The list is the
// ascending union of the windows' bins; a window that starts below bin 0 is indexed by the reference with negative k, which
// Python wraps to the top of the spectrum (x_dft[k], k < 0 -> x_dft[nbins + k]): such a window gets a run of its own after
// the ascending part (bins nbins + k0 .. nbins - 1, then 0 .. k1 - 1: the reference's order).  A window that reaches past
// the last bin, or below -nbins, raises IndexError in the reference: MPX_EINVAL here.
struct HeWindows {
    std::vector<int> k0, k1;     // bin numbers as the reference computes them (k0 may be negative)
    std::vector<int> c0, c1;     // the window's run of `bins`
    std::vector<double> w;       // 1 / harmonic
    std::vector<int> bins;       // compact bin list (never empty)
    int kmin = 0, kmax = 0;      // [kmin, kmax) covers every listed bin
    bool wrapped = false;
    int sorted_bins = 0;         // length of the ascending part
};
static int he_windows(mpx_ctx* ctx, int fs, int N, const mpx_he_params& p, HeWindows& W) {
    const int nbins = N / 2 + 1;
    // note table: librosa.cqt_frequencies(12, fmin=note_to_hz('C3')) harmonic_energy.py:33
    const double c3 = 440.0 * std::pow(2.0, (48.0 - 69.0) / 12.0);
    const double divisor_ratio = (fs / 4.0) / N;  // harmonic_energy.py:35 (quirk A.2)
    int kmin = 1 << 30, kmax = -(1 << 30);
    for (int n = 0; n < 12; ++n) {
        const double note = c3 * std::pow(2.0, n / 12.0);
        for (int oct = 1; oct <= p.num_octave; ++oct)
            for (int h = 1; h <= p.num_harmonic; ++h) {
                const double kp = he_round_half_even((note * oct * h) / divisor_ratio);
                const int a = (int)(kp - p.num_bins * h), b = (int)(kp + p.num_bins * h);
                W.k0.push_back(a);
                W.k1.push_back(b);
                W.w.push_back(1.0 / h);
                if (b > a) {
                    if (a < -nbins || b > nbins)
                        return set_error(ctx, MPX_EINVAL, "harmonic-energy window [%d,%d) outside the %d-bin spectrum (the "
                                         "reference raises IndexError here)", a, b, nbins);
                    if (a < 0) W.wrapped = true;
                    kmin = a < kmin ? a : kmin;
                    kmax = b > kmax ? b : kmax;
                }
            }
    }
    if (kmin > kmax) kmin = kmax = 0;
    if (W.wrapped) {   // bins at both ends of the spectrum are looked at
        kmin = 0;
        kmax = nbins;
    }
    W.kmin = kmin;
    W.kmax = kmax;
    for (size_t w = 0; w < W.k0.size(); ++w)
        if (W.k0[w] >= 0)
            for (int k = W.k0[w]; k < W.k1[w]; ++k) W.bins.push_back(k);
    std::sort(W.bins.begin(), W.bins.end());
    W.bins.erase(std::unique(W.bins.begin(), W.bins.end()), W.bins.end());
    W.sorted_bins = (int)W.bins.size();
    W.c0.resize(W.k0.size());
    W.c1.resize(W.k0.size());
    for (size_t w = 0; w < W.k0.size(); ++w) {
        const int len = W.k1[w] > W.k0[w] ? W.k1[w] - W.k0[w] : 0;
        int c0;
        if (len && W.k0[w] < 0) {
            c0 = (int)W.bins.size();
            for (int k = W.k0[w]; k < W.k1[w]; ++k) W.bins.push_back(k < 0 ? k + nbins : k);
        } else {
            c0 = (int)(std::lower_bound(W.bins.begin(), W.bins.begin() + W.sorted_bins, W.k0[w]) - W.bins.begin());
        }
        W.c0[w] = c0;
        W.c1[w] = c0 + len;
    }
    if (W.bins.empty()) W.bins.push_back(0);
    return MPX_OK;
}

template <typename Real>
static int he_build_plan(mpx_ctx* ctx, int fs, int N, const mpx_he_params& p, HePlan& plan) {
    const int M = N / 2;
    HeWindows W;
    if (int rc = he_windows(ctx, fs, N, p, W)) return rc;
    std::vector<int> k0 = W.k0, k1 = W.k1;
    std::vector<Real> ww(W.w.begin(), W.w.end());
    const int kmin = W.kmin, kmax = W.kmax;
    plan.wrapped = W.wrapped;
    plan.nwin = (int)k0.size();
    plan.h_k0 = k0;   // (bin numbers; the device arrays below index the compact bin list)
    plan.h_k1 = k1;
    plan.wins_per_note = p.num_octave * p.num_harmonic;
    plan.num_harmonic = p.num_harmonic;
    plan.kmin = kmin;
    plan.kmax = kmax;
    // scipy.signal.hamming(N) (symmetric) angles: cos/sin(2 pi s/(N-1)) for every s; the kernel
    // combines a per-thread base angle with a per-register offset angle
    std::vector<cx<Real>> wang(N);
    for (int i = 0; i < N; ++i) {
        const long double ang = 2.0L * M_PIl * i / (long double)(N - 1);
        wang[i] = {(Real)cosl(ang), (Real)sinl(ang)};
    }
    std::vector<cx<Real>> tw(M), twn(M + 1);
    for (int j = 0; j < M; ++j) {
        const long double ang = -2.0L * M_PIl * j / (long double)M;
        tw[j] = {(Real)cosl(ang), (Real)sinl(ang)};
    }
    for (int k = 0; k <= M; ++k) {
        const long double ang = -2.0L * M_PIl * k / (long double)N;
        twn[k] = {(Real)cosl(ang), (Real)sinl(ang)};
    }
    plan.window = upload(ctx, wang.data(), wang.size() * sizeof(cx<Real>));
    plan.tw = upload(ctx, tw.data(), tw.size() * sizeof(cx<Real>));
    plan.twn = upload(ctx, twn.data(), twn.size() * sizeof(cx<Real>));
    // the compact bin list (he_windows): ascending bins, a window is a run of it
    std::vector<int> bins = W.bins;
    k0 = W.c0;
    k1 = W.c1;
    plan.nb = (int)bins.size();
    plan.h_bins = bins;
    std::vector<cx<Real>> twnb(bins.size());
    for (size_t i = 0; i < bins.size(); ++i) twnb[i] = twn[bins[i] <= M ? bins[i] : 0];
    plan.twnb = upload(ctx, twnb.data(), twnb.size() * sizeof(cx<Real>));
    if (!plan.twnb) return MPX_ENOMEM;
    plan.wk0 = (int*)upload(ctx, k0.data(), k0.size() * sizeof(int));
    plan.wk1 = (int*)upload(ctx, k1.data(), k1.size() * sizeof(int));
    plan.ww = upload(ctx, ww.data(), ww.size() * sizeof(Real));
    if (!plan.window || !plan.tw || !plan.twn || !plan.wk0 || !plan.wk1 || !plan.ww) return MPX_ENOMEM;
    return MPX_OK;
}

// The wave-per-frame kernel (mpx_he_wave.hpp): 4096-sample frames (the headline shape) and, two passes per frame, the
// reference's default 8192; fp64, at most 256 window bins.
constexpr int HEW_WAVES = 8, HEW_ROUNDS = 4;
constexpr int HEW_WAVES2 = 7;   // 8192-sample frames: the window table is 32 KB, seven transpose buffers fit next to it
constexpr int HEW_WAVES2P = 6;  // ... or three PAIRS of waves, a frame per pair, and 4 KB per pair for the spectrum pass 0 hands over
static bool he_wave_applies(const mpx_ctx* ctx, const HePlan& plan) {
    if (ctx->he_kernel == MPX_HE_KERNEL_WORKGROUP) return false;   // mpx_set_option: the workgroup-per-frame kernel below
    if (plan.wrapped) return false;   // windows that wrap to the top of the spectrum: rows the pruned bin copy does not hold
    return plan.nb <= 64 * HEW_ROUNDS && plan.nwin <= 192;
}
template <int HALVES>
static int he_wave_launch(mpx_ctx* ctx, const HePlan& plan, const float* d_signal, int64_t n, const FrameDesc* d_desc,
                          int64_t num_frames, int hop, double* d_out, double* d_sum, hipStream_t stream) {
    constexpr int N = 4096 * HALVES;
    // 8192-sample frames: both passes of a frame on ONE wave (seven per CU) by default; MPX_OPT_HE_KERNEL =
    // MPX_HE_KERNEL_WAVE_PAIRS gives a frame to TWO waves (three pairs per CU), one per pass -- same bits.  Measured (round 5,
    // 8196 frames from two 268 MB inputs in turn, scripts/dev/he8192_ab.py, he8192_pmc.py): pairs fetch 1.2 x the compulsory
    // bytes where one wave fetches 2.2 x (every line twice), but take 121-126 us against 122 (106 against 88 with the input
    // resident in L2: six waves per CU instead of seven) -- the second fetch is not what the streamed case waits for.
    // Round 6: AUTO picks by where the samples will come from.  A batch whose samples do not fit the chip's L2 (32 MB over the
    // eight XCDs) is streamed, and there the pairs cost nothing in time and fetch every line once: they are the default for
    // such a call (he_default_8192 of bench.py: 268 MB); a batch that fits stays on one wave per frame (88 us against 106
    // resident).  The choice looks at the call's own size only and the two arrangements return the same bits.
    constexpr long long HEW_PAIRS_FROM_BYTES = 32LL << 20;
    const long long touched = num_frames * (long long)(hop < N ? hop : N) * (long long)sizeof(float);
    const bool paired = HALVES == 2 && (ctx->he_kernel == MPX_HE_KERNEL_WAVE_PAIRS ||
                                        (ctx->he_kernel == MPX_HE_KERNEL_AUTO && touched >= HEW_PAIRS_FROM_BYTES));
    const int WAVES = HALVES == 1 ? HEW_WAVES : (paired ? HEW_WAVES2P : HEW_WAVES2);
    if (!plan.whalf) {
        // scipy.signal.hamming(N), harmonic_energy.py:42, as the kernel reads it: [pass h][pair pm < 1024] = the window at the
        // samples 2 HALVES pm + 2 h and the next one (the upper half of a pass is the mirror image of pass HALVES - 1 - h)
        std::vector<double> wh((size_t)2048 * HALVES);
        for (int h = 0; h < HALVES; ++h)
            for (int pm = 0; pm < 1024; ++pm)
                for (int j = 0; j < 2; ++j) {
                    const int i = 2 * HALVES * pm + 2 * h + j;
                    wh[(size_t)h * 2048 + 2 * pm + j] = (double)(0.54L - 0.46L * cosl(2.0L * M_PIl * i / (long double)(N - 1)));
                }
        std::vector<unsigned> sl(2 * plan.h_bins.size());
        for (size_t i = 0; i < plan.h_bins.size(); ++i) {
            const int kp = plan.h_bins[i] & 1023, km = (1024 - kp) & 1023;
            sl[2 * i] = (unsigned)hw_slot(0, kp) | ((unsigned)hw_slot(0, km) << 16);
            sl[2 * i + 1] = (unsigned)hw_slot(1, kp) | ((unsigned)hw_slot(1, km) << 16);
        }
        plan.wslots = (unsigned*)upload(ctx, sl.data(), sl.size() * sizeof(unsigned));
        plan.whalf = upload(ctx, wh.data(), wh.size() * sizeof(double));
        if (!plan.whalf || !plan.wslots) return MPX_ENOMEM;
        // one window per lane and one note per quad needs the reference's 12 x 2 x 2 windows, none empty, none wider than 8
        bool quad = plan.nwin == 48 && plan.wins_per_note == 4 && plan.num_harmonic == 2;
        for (size_t w = 0; quad && w < plan.h_k0.size(); ++w) quad = plan.h_k1[w] > plan.h_k0[w] && plan.h_k1[w] - plan.h_k0[w] <= 8;
        plan.quad_tail = quad ? 1 : 0;
    }
    HeWaveArgs a;
    a.sig = d_signal;
    a.n = n;
    a.desc = d_desc;
    a.num_frames = num_frames;
    a.hop = hop;
    a.whalf = (const double*)plan.whalf;
    a.tw = (const cx<double>*)plan.tw;
    a.wk0 = plan.wk0;
    a.wk1 = plan.wk1;
    a.ww = (const double*)plan.ww;
    a.slots = plan.wslots;
    a.twnb = (const cx<double>*)plan.twnb;   // W_N^k
    a.escratch = nullptr;
    a.nb = plan.nb;
    a.nwin = plan.nwin;
    a.wins_per_note = plan.wins_per_note;
    a.num_harmonic = plan.num_harmonic;
    a.quad_tail = plan.quad_tail;
    if (!d_out) {   // the sum over frames goes through the rows: which wave computes a frame is decided at run time
        int rc = ensure(ctx, ctx->d_frames_out, (size_t)num_frames * 12 * sizeof(double));
        if (rc) return rc;
        d_out = (double*)ctx->d_frames_out.p;
    }
    a.out = d_out;
    a.partial = nullptr;
    // one workgroup per CU, each owning a contiguous run of frames
    long long g = ctx->num_cus < num_frames ? ctx->num_cus : num_frames;
    const long long per = (num_frames + g - 1) / g;
    g = (num_frames + per - 1) / per;
    if (d_sum) {
        int rc = ensure(ctx, ctx->d_partials, (size_t)g * 12 * sizeof(double));
        if (rc) return rc;
        a.partial = (double*)ctx->d_partials.p;
    }
    if (HALVES == 2 && !paired) {   // pass 0's spectrum at the window bins, 4 KB per wave (d_ws4 is the ESACF path's: not in use here)
        int rc = ensure(ctx, ctx->d_ws4, (size_t)g * WAVES * 64 * HEW_ROUNDS * sizeof(cx<double>));
        if (rc) return rc;
        a.escratch = (cx<double>*)ctx->d_ws4.p;
    }
    const size_t lds = (size_t)hw_shared_bytes(HEW_ROUNDS, plan.nwin, HALVES) + (size_t)WAVES * HW_XBUF +
                       (paired ? (size_t)(WAVES / 2) * 64 * HEW_ROUNDS * sizeof(cx<double>) + 64 : 0);
    // every frame whole, inside the signal and (float2 loads of the 4096 kernel) 8-byte aligned: the instantiation without the
    // ragged loader (no scratch)
    const bool all_fast = !d_desc && (hop & 1) == 0 && (reinterpret_cast<uintptr_t>(d_signal) & 7) == 0 &&
                          (num_frames - 1) * (long long)hop + N <= n;
    using kern_t = void (*)(HeWaveArgs, cx<double>*);
    kern_t kern;
    const char* okey;
    if constexpr (HALVES == 1) {
        // rows of ZA / ZB the window bins (and their mirrors) live in: the 44.1 kHz instantiation when they fit its 22 rows
        unsigned k2 = 0;
        for (int k : plan.h_bins) k2 |= hw_k2_bits(k);
        const bool k44 = (k2 & ~HW_K2_44K) == 0;
        static const kern_t kerns[4] = {he_wave_kernel<HEW_WAVES, HEW_ROUNDS, false, false, HW_K2_ALL>,
                                        he_wave_kernel<HEW_WAVES, HEW_ROUNDS, false, true, HW_K2_ALL>,
                                        he_wave_kernel<HEW_WAVES, HEW_ROUNDS, false, false, HW_K2_44K>,
                                        he_wave_kernel<HEW_WAVES, HEW_ROUNDS, false, true, HW_K2_44K>};
        kern = kerns[(k44 ? 2 : 0) + (all_fast ? 1 : 0)];
        okey = "he_wave_lds";
        if (!ctx->occupancy.count(okey))
            for (kern_t k : kerns) MPX_HIP(ctx, hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    } else {
        static const kern_t kerns[4] = {he_wave_kernel<HEW_WAVES2, HEW_ROUNDS, false, false, HW_K2_ALL, 2>,
                                        he_wave_kernel<HEW_WAVES2, HEW_ROUNDS, false, true, HW_K2_ALL, 2>,
                                        he_wave_kernel<HEW_WAVES2P, HEW_ROUNDS, false, false, HW_K2_ALL, 2, true>,
                                        he_wave_kernel<HEW_WAVES2P, HEW_ROUNDS, false, true, HW_K2_ALL, 2, true>};
        kern = kerns[(paired ? 2 : 0) + (all_fast ? 1 : 0)];
        okey = "he_wave2_lds";
        if (!ctx->occupancy.count(okey))
            for (kern_t k : kerns) MPX_HIP(ctx, hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    ctx->occupancy[okey] = 1;
    if (lds > 160 * 1024) return set_error(ctx, MPX_EUNSUPPORTED, "harmonic energy: the wave kernel needs %zu B of LDS", lds);
    prof_mark(ctx, stream, "he_wave_kernel");
    hipLaunchKernelGGL(kern, dim3((unsigned)g), dim3(WAVES * 64), lds, stream, a, (cx<double>*)nullptr);
    prof_mark(ctx, stream, nullptr);
    MPX_HIP(ctx, hipGetLastError());
    if (d_sum) {
        prof_mark(ctx, stream, "sum_all_kernel");
        hipLaunchKernelGGL(sum_all_kernel, dim3(1), dim3(256), 0, stream, (const double*)a.partial, g, d_sum);
        prof_mark(ctx, stream, nullptr);
        MPX_HIP(ctx, hipGetLastError());
    }
    return MPX_OK;
}

template <int N, int T, typename Real>
static int he_launch(mpx_ctx* ctx, const HePlan& plan, const float* d_signal, int64_t n,
                     const FrameDesc* d_desc, int64_t num_frames, int hop, double* d_out, double* d_sum,
                     hipStream_t stream) {
    if constexpr ((N == 4096 || N == 8192) && std::is_same<Real, double>::value) {
        if (he_wave_applies(ctx, plan))
            return he_wave_launch<N / 4096>(ctx, plan, d_signal, n, d_desc, num_frames, hop, d_out, d_sum, stream);
    }
    HeArgs<Real> a;
    a.sig = d_signal;
    a.n = n;
    a.desc = d_desc;
    a.num_frames = num_frames;
    a.hop = hop;
    a.wbase = (const cx<Real>*)plan.window;  // angles of samples 0..2T-1
    {
        // offset angles of the EPT register slots of thread 0 (first-pass read order)
        constexpr int M_ = N / 2, EPT_ = M_ / T;
        constexpr int R0 = Plan<M_, EPT_>::radix(0), NB = M_ / R0;
        auto key = std::string("he_woffs_") + std::to_string(N) + "_" + std::to_string(T) + (sizeof(Real) == 4 ? "f" : "d");
        auto it = ctx->misc_plans.find(key);
        if (it == ctx->misc_plans.end()) {
            std::vector<cx<Real>> offs(EPT_);
            for (int e = 0; e < EPT_; ++e) {
                const int b = e / R0, r = e % R0;
                const long double ang = 2.0L * M_PIl * (2.0L * (b * T + r * NB)) / (long double)(N - 1);
                offs[e] = {(Real)(-0.46L * cosl(ang)), (Real)(-0.46L * sinl(ang))};  // -0.46 (cos, sin): see he_kernel
            }
            void* d = upload(ctx, offs.data(), offs.size() * sizeof(cx<Real>));
            if (!d) return MPX_ENOMEM;
            it = ctx->misc_plans.emplace(key, std::vector<void*>{d}).first;
        }
        a.woffs = (const cx<Real>*)it->second[0];
    }
    a.tw = (const cx<Real>*)plan.tw;
    a.twn = (const cx<Real>*)plan.twn;
    a.wk0 = plan.wk0;
    a.wk1 = plan.wk1;
    if (!plan.slots) {
        constexpr int M_ = N / 2;
        std::vector<unsigned> sl(plan.h_bins.size());
        for (size_t i = 0; i < sl.size(); ++i) {
            const int k = plan.h_bins[i];
            const int ka = k & (M_ - 1), kb = (M_ - k) & (M_ - 1);
            unsigned sa, sb;
            if constexpr (he_uses_dif<N, T>()) {
                sa = (unsigned)sigma<M_>(dif_pos<M_>(ka));
                sb = (unsigned)sigma<M_>(dif_pos<M_>(kb));
            } else {
                sa = (unsigned)lds_slot(ka);
                sb = (unsigned)lds_slot(kb);
            }
            sl[i] = sa | (sb << 16);
        }
        plan.slots = (unsigned*)upload(ctx, sl.data(), sl.size() * sizeof(unsigned));
        if (!plan.slots) return MPX_ENOMEM;
    }
    a.slots = plan.slots;
    a.twnb = (const cx<Real>*)plan.twnb;
    a.nb = plan.nb;
    a.ww = (const Real*)plan.ww;
    a.nwin = plan.nwin;
    a.wins_per_note = plan.wins_per_note;
    a.num_harmonic = plan.num_harmonic;
    a.kmin = plan.kmin;
    a.kmax = plan.kmax;
    a.out = d_out;
    a.partial = nullptr;
    a.sum = nullptr;
    a.counter = nullptr;
    const size_t lds = sizeof(cx<Real>) * he_buf_slots<N, T>() + sizeof(Real) * (size_t)(plan.nb + 2 * plan.nwin) + 8 * (size_t)plan.nwin + 4 * (size_t)plan.nb + 48 + sizeof(cx<Real>) * (N / 2 / T);
    if (lds > 160 * 1024)
        return set_error(ctx, MPX_EUNSUPPORTED, "frame %d needs %zu B of LDS (> 160 KiB)", N, lds);
    auto kern = he_kernel<N, T, Real>;
    // persistent grid: as many workgroups as the chip holds at once, each owning a contiguous frame range
    // (occupancy query and LDS opt-in are done once per kernel/LDS size and cached: they cost host time)
    const std::string okey = "he_occ_" + std::to_string(N) + "_" + std::to_string(T) + "_" + std::to_string(sizeof(Real)) +
                             "_" + std::to_string(lds);
    auto oit = ctx->occupancy.find(okey);
    if (oit == ctx->occupancy.end()) {
        if (lds > 48 * 1024)
            MPX_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int o = 0;
        MPX_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, (const void*)kern, T, lds));
        oit = ctx->occupancy.emplace(okey, o < 1 ? 1 : o).first;
    }
    const int occ = oit->second;
    long long g = (long long)occ * ctx->num_cus * dev_env_int("MPX_HE_OVERSUB", 1);
    if (g > num_frames) g = num_frames;
    const long long per = (num_frames + g - 1) / g;
    g = (num_frames + per - 1) / per;
    if (d_sum) {
        const size_t groups = (size_t)(g + 31) / 32;
        int rc = ensure(ctx, ctx->d_partials, ((size_t)g + groups) * 12 * sizeof(double));
        if (rc) return rc;
        if (ctx->d_counter.bytes < (groups + 1) * sizeof(unsigned)) {
            if ((rc = ensure(ctx, ctx->d_counter, (groups + 1) * sizeof(unsigned) + 4096))) return rc;
            MPX_HIP(ctx, hipMemsetAsync(ctx->d_counter.p, 0, ctx->d_counter.bytes, stream));
        }
        a.partial = (double*)ctx->d_partials.p;
        a.sum = d_sum;
        a.counter = (unsigned*)ctx->d_counter.p;
    }
    prof_mark(ctx, stream, "he_kernel");
    hipLaunchKernelGGL(kern, dim3((unsigned)g), dim3(T), lds, stream, a);
    prof_mark(ctx, stream, nullptr);
    MPX_HIP(ctx, hipGetLastError());
    return MPX_OK;
}

template <typename Real>
static int he_dispatch(mpx_ctx* ctx, const HePlan& plan, const float* d_signal, int64_t n,
                       const FrameDesc* d_desc, int64_t num_frames, int frame, int hop, double* d_out,
                       double* d_sum, hipStream_t stream) {
    switch (frame) {
        case 1024: return he_launch<1024, 64, Real>(ctx, plan, d_signal, n, d_desc, num_frames, hop, d_out, d_sum, stream);
        case 2048: return he_launch<2048, 128, Real>(ctx, plan, d_signal, n, d_desc, num_frames, hop, d_out, d_sum, stream);
        case 4096: return he_launch<4096, HE4096_T, Real>(ctx, plan, d_signal, n, d_desc, num_frames, hop, d_out, d_sum, stream);
        case 8192: return he_launch<8192, 512, Real>(ctx, plan, d_signal, n, d_desc, num_frames, hop, d_out, d_sum, stream);
        case 16384: return he_launch<16384, 512, Real>(ctx, plan, d_signal, n, d_desc, num_frames, hop, d_out, d_sum, stream);
        default:
            return set_error(ctx, MPX_EUNSUPPORTED,
                             "harmonic energy: frame size %d is not a power of two in [1024, 16384]", frame);
    }
}

// ------------------------------------------------------------------ arbitrary frame lengths
// Frame sizes that are not a power of two in [1024, 16384] (the reference takes any `frame_size`): the N-point DFT of
// the windowed frame as a chirp-z transform on the padded Stockham engine, one workgroup per frame, then the same
// window maxima / pitch-class sums.  Complete and exact, not tuned (the power-of-two kernel above is the fast path).
struct HeBlueArgs {
    const float* sig;
    long long n;
    const FrameDesc* desc;
    long long num_frames;
    int N, hop;
    const double* win;       // [N] symmetric Hamming (scipy.signal.hamming, harmonic_energy.py:42)
    const cx<double>* chirp; // [N] exp(i pi n^2 / N)
    const cx<double>* bhat;  // [L] FFT_L(chirp filter) / L
    const cx<double>* tw;    // [L] W_L
    const int* bins;         // [nb]
    const int* wk0;
    const int* wk1;
    const double* ww;
    int nb, nwin, wins_per_note, num_harmonic;
    double* out;             // [F, 12]
    int* argmax;             // debug tap (mpx_harmonic_energy_argmax), else NULL: [F, nwin] bin of each window's FIRST maximum,
    const int* wbase;        //   counted like the reference does (wbase[w] = k0, negative in a wrapped window); INT_MIN: empty
};

template <int L, int T>
__global__ __launch_bounds__(T) void he_blue_kernel(HeBlueArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cx<double>* buf = reinterpret_cast<cx<double>*>(smem);
    double* mag = reinterpret_cast<double*>(smem + sizeof(cx<double>) * lds_slots(L));  // [nb] then winmax [nwin]
    double* winmax = mag + a.nb;
    const int tid = threadIdx.x, N = a.N;
    const long long f = blockIdx.x;
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
    cx<double> regs[L / T];
    for (int n = tid; n < L; n += T) {
        cx<double> v = {0.0, 0.0};
        if (n < N) {
            const double s = (n < valid ? (double)x[n] : 0.0) * a.win[n];
            const cx<double> ch = a.chirp[n];
            v = {s * ch.x, -s * ch.y};  // s * conj(chirp)
        }
        buf[lds_slot(n)] = v;
    }
    __syncthreads();
    fft_lds<L, T, false, double>(buf, a.tw, regs, tid);
    for (int k = tid; k < L; k += T) {
        const cx<double> p = cmul(buf[lds_slot(k)], a.bhat[k]);
        buf[lds_slot(k)] = {p.y, p.x};  // swapped: the next forward FFT acts as the inverse
    }
    __syncthreads();
    fft_lds<L, T, false, double>(buf, a.tw, regs, tid);
    for (int i = tid; i < a.nb; i += T) {
        const int k = a.bins[i];  // k <= N/2
        const cx<double> b = buf[lds_slot(k)];
        const cx<double> ch = a.chirp[k < N ? k : 0];
        const cx<double> z = cmul(cx<double>{b.y, b.x}, cx<double>{ch.x, -ch.y});
        mag[i] = sqrt(sqrt(z.x * z.x + z.y * z.y));  // sqrt(|rfft|), harmonic_energy.py:43
    }
    __syncthreads();
    for (int wi = tid; wi < a.nwin; wi += T) {
        double m = -INFINITY;
        int best = -1;
        for (int k = a.wk0[wi]; k < a.wk1[wi]; ++k)
            if (mag[k] > m) {   // strict: the first maximum wins (harmonic_energy.py:60-62)
                m = mag[k];
                best = k;
            }
        winmax[wi] = m;
        if (a.argmax) a.argmax[f * a.nwin + wi] = best < 0 ? INT_MIN : a.wbase[wi] + (best - a.wk0[wi]);
    }
    __syncthreads();
    if (tid < 12) {
        double chroma = 0.0;
        const int base = tid * a.wins_per_note;
        for (int o = 0; o < a.wins_per_note; o += a.num_harmonic) {
            double note_sum = 0.0;
            for (int h = 0; h < a.num_harmonic; ++h) note_sum += winmax[base + o + h] * a.ww[base + o + h];
            chroma += note_sum;
        }
        a.out[f * 12 + tid] = chroma;
    }
}

// Frames whose chirp-z does not fit 8192 points (N + highest window bin > 8192: non-powers of two above ~6900 samples,
// anything above 16384): the input is decimated by R -- X[k] = sum_r W_N^(r k) G_r[k], G_r[k] = sum_m s[R m + r] W_N^(R m k) --
// and every G_r is a chirp-z transform of ceil(N / R) points with the chirp exp(i pi R j^2 / N) (a zoom: only the K bins below
// the highest window are evaluated), R passes through the same 8192-point engine; a thread accumulates its (at most two) window
// bins over the passes, the factors W_N^(r k) conj(chirp[k]) come from a table built with the plan.
struct HeBlueSplitArgs {
    HeBlueArgs b;            // chirp: [ceil(N/R)] exp(i pi R j^2 / N); bhat: filter spectrum of that chirp
    int R, n1;               // decimation, ceil(N / R)
    const cx<double>* coef;  // [R][nb] W_N^(r k) conj(chirp[k]) at the window bins
};

template <int L, int T, int PB>   // PB: window bins per thread (nb <= PB * T, checked by the host)
__global__ __launch_bounds__(T) void he_blue_split_kernel(HeBlueSplitArgs sa) {
    const HeBlueArgs& a = sa.b;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cx<double>* buf = reinterpret_cast<cx<double>*>(smem);
    double* mag = reinterpret_cast<double*>(smem + sizeof(cx<double>) * lds_slots(L));  // [nb] then winmax [nwin]
    double* winmax = mag + a.nb;
    const int tid = threadIdx.x, N = a.N, R = sa.R, n1 = sa.n1;
    const long long f = blockIdx.x;
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
    cx<double> regs[L / T];
    cx<double> acc[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) acc[j] = {0.0, 0.0};
    for (int r = 0; r < R; ++r) {
        for (int m = tid; m < L; m += T) {
            cx<double> v = {0.0, 0.0};
            const long long n = (long long)R * m + r;
            if (m < n1 && n < N) {
                const double s = (n < valid ? (double)x[n] : 0.0) * a.win[n];
                const cx<double> ch = a.chirp[m];
                v = {s * ch.x, -s * ch.y};  // s * conj(chirp)
            }
            buf[lds_slot(m)] = v;
        }
        __syncthreads();
        fft_lds<L, T, false, double>(buf, a.tw, regs, tid);
        for (int k = tid; k < L; k += T) {
            const cx<double> p = cmul(buf[lds_slot(k)], a.bhat[k]);
            buf[lds_slot(k)] = {p.y, p.x};  // swapped: the next forward FFT acts as the inverse
        }
        __syncthreads();
        fft_lds<L, T, false, double>(buf, a.tw, regs, tid);
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const int i = tid + j * T;
            if (i < a.nb) {
                const cx<double> b = buf[lds_slot(a.bins[i])];
                const cx<double> z = cmul(cx<double>{b.y, b.x}, sa.coef[(size_t)r * a.nb + i]);
                acc[j] = {acc[j].x + z.x, acc[j].y + z.y};
            }
        }
        __syncthreads();   // the next pass overwrites buf
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int i = tid + j * T;
        if (i < a.nb) mag[i] = sqrt(sqrt(acc[j].x * acc[j].x + acc[j].y * acc[j].y));  // sqrt(|rfft|), harmonic_energy.py:43
    }
    __syncthreads();
    for (int wi = tid; wi < a.nwin; wi += T) {
        double m = -INFINITY;
        int best = -1;
        for (int k = a.wk0[wi]; k < a.wk1[wi]; ++k)
            if (mag[k] > m) {   // strict: the first maximum wins (harmonic_energy.py:60-62)
                m = mag[k];
                best = k;
            }
        winmax[wi] = m;
        if (a.argmax) a.argmax[f * a.nwin + wi] = best < 0 ? INT_MIN : a.wbase[wi] + (best - a.wk0[wi]);
    }
    __syncthreads();
    if (tid < 12) {
        double chroma = 0.0;
        const int base = tid * a.wins_per_note;
        for (int o = 0; o < a.wins_per_note; o += a.num_harmonic) {
            double note_sum = 0.0;
            for (int h = 0; h < a.num_harmonic; ++h) note_sum += winmax[base + o + h] * a.ww[base + o + h];
            chroma += note_sum;
        }
        a.out[f * 12 + tid] = chroma;
    }
}

static void he_host_fft(std::vector<cx<double>>& a) {  // in-place radix-2, forward; plan tables only
    const size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const long double ang = -2.0L * M_PIl / (long double)len;
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const cx<double> w = {(double)cosl(ang * k), (double)sinl(ang * k)};
                const cx<double> t = a[i + k + len / 2];
                const cx<double> u = a[i + k], v = {t.x * w.x - t.y * w.y, t.x * w.y + t.y * w.x};
                a[i + k] = {u.x + v.x, u.y + v.y};
                a[i + k + len / 2] = {u.x - v.x, u.y - v.y};
            }
    }
}

static int he_blue_run(mpx_ctx* ctx, const float* d_signal, int64_t n, const FrameDesc* d_desc, int64_t num_frames, int fs,
                       const mpx_he_params& p, int N, int hop, double* d_rows, double* d_sum, hipStream_t stream,
                       int* d_argmax = nullptr) {
    char keyb[128];
    snprintf(keyb, sizeof keyb, "he_blue_%d_%d_%d_%d_%d", fs, N, p.num_harmonic, p.num_octave, p.num_bins);
    const std::string key = keyb;
    auto it = ctx->misc_plans.find(key);
    if (it == ctx->misc_plans.end()) {
        HeWindows W;
        if (int rc = he_windows(ctx, fs, N, p, W)) return rc;
        std::vector<int> k0 = W.c0, k1 = W.c1, bins = W.bins;
        std::vector<double> ww = W.w;
        // Only the bins below the highest window are looked at: the chirp-z convolution spans chirp[-(N-1) .. K-1],
        // K = that bin + 1, so a circular length of N + K - 1 does (not 2N - 1): sizes up to ~6900 samples fit the
        // 8192-point engine with the default windows, and 2049..3500 take the 4096-point one.
        const int K = *std::max_element(bins.begin(), bins.end()) + 1;   // (a wrapped window reaches bin N / 2)
        int L = 1024;
        while (L < N + K - 1 && L < 8192) L <<= 1;
        // beyond 8192 points: decimate the input by R (he_blue_split_kernel), R passes of ceil(N / R) + K - 1 <= 8192 points
        int R = 1;
        while ((N + R - 1) / R + K - 1 > 8192 && R < 64) ++R;
        if ((N + R - 1) / R + K - 1 > 8192 || (R > 1 && bins.size() > 4096))
            return set_error(ctx, MPX_EUNSUPPORTED, "harmonic energy: frame size %d with %zu window bins up to bin %d does not fit "
                             "64 passes of the 8192-point chirp-z", N, bins.size(), K - 1);
        const int n1 = (N + R - 1) / R;
        const int J = n1 > K ? n1 : K;   // chirp samples needed: inputs m < n1, outputs k < K
        std::vector<double> win(N);
        for (int i = 0; i < N; ++i) win[i] = N == 1 ? 1.0 : 0.54 - 0.46 * std::cos(2.0 * M_PI * i / (double)(N - 1));
        std::vector<cx<double>> chirp(J), filt(L, cx<double>{0.0, 0.0}), tw(L);
        for (long long i = 0; i < J; ++i) {   // exp(i pi R j^2 / N), the phase reduced exactly
            const long long q = (long long)(((unsigned long long)R * (unsigned long long)(i * i)) % (unsigned long long)(2LL * N));
            const long double ang = M_PIl * (long double)q / (long double)N;
            chirp[i] = {(double)cosl(ang), (double)sinl(ang)};
        }
        for (int m = 0; m < K && m < J; ++m) filt[m] = chirp[m];     // chirp[k - n], k - n = 0 .. K-1
        for (int m = 1; m < n1; ++m) filt[L - m] = chirp[m];         // k - n = -1 .. -(n1-1) (the chirp is even)
        he_host_fft(filt);
        for (auto& v : filt) {
            v.x /= L;
            v.y /= L;
        }
        for (int j = 0; j < L; ++j) {
            const long double ang = -2.0L * M_PIl * j / (long double)L;
            tw[j] = {(double)cosl(ang), (double)sinl(ang)};
        }
        std::vector<int> meta = {(int)bins.size(), (int)k0.size(), p.num_octave * p.num_harmonic, p.num_harmonic, L, R, n1};
        std::vector<cx<double>> coef;   // R > 1: W_N^(r k) conj(chirp[k]) per pass and window bin
        if (R > 1) {
            coef.resize((size_t)R * bins.size());
            for (int r = 0; r < R; ++r)
                for (size_t i = 0; i < bins.size(); ++i) {
                    const long long k = bins[i];
                    const long double ang = -2.0L * M_PIl * (long double)((r * k) % N) / (long double)N;
                    const cx<double> w = {(double)cosl(ang), (double)sinl(ang)}, c = chirp[(size_t)k];
                    coef[(size_t)r * bins.size() + i] = {w.x * c.x + w.y * c.y, w.y * c.x - w.x * c.y};   // w * conj(c)
                }
        } else {
            coef.resize(1);
        }
        std::vector<void*> d = {upload(ctx, win.data(), win.size() * sizeof(double)),
                                upload(ctx, chirp.data(), chirp.size() * sizeof(cx<double>)),
                                upload(ctx, filt.data(), filt.size() * sizeof(cx<double>)),
                                upload(ctx, tw.data(), tw.size() * sizeof(cx<double>)),
                                upload(ctx, bins.data(), bins.size() * sizeof(int)),
                                upload(ctx, k0.data(), k0.size() * sizeof(int)),
                                upload(ctx, k1.data(), k1.size() * sizeof(int)),
                                upload(ctx, ww.data(), ww.size() * sizeof(double)),
                                upload(ctx, coef.data(), coef.size() * sizeof(cx<double>)),
                                upload(ctx, W.k0.data(), W.k0.size() * sizeof(int))};
        for (void* q : d)
            if (!q) return MPX_ENOMEM;
        std::vector<unsigned char> blob(meta.size() * sizeof(int));
        std::memcpy(blob.data(), meta.data(), blob.size());
        ctx->host_blobs[key] = std::move(blob);
        it = ctx->misc_plans.emplace(key, d).first;
    }
    const int* meta = reinterpret_cast<const int*>(ctx->host_blobs[key].data());
    double* rows = d_rows;
    if (!rows) {
        int rc = ensure(ctx, ctx->d_frames_out, (size_t)(num_frames ? num_frames : 1) * 12 * sizeof(double));
        if (rc) return rc;
        rows = (double*)ctx->d_frames_out.p;
    }
    HeBlueArgs a;
    a.sig = d_signal;
    a.n = n;
    a.desc = d_desc;
    a.num_frames = num_frames;
    a.N = N;
    a.hop = hop;
    a.win = (const double*)it->second[0];
    a.chirp = (const cx<double>*)it->second[1];
    a.bhat = (const cx<double>*)it->second[2];
    a.tw = (const cx<double>*)it->second[3];
    a.bins = (const int*)it->second[4];
    a.wk0 = (const int*)it->second[5];
    a.wk1 = (const int*)it->second[6];
    a.ww = (const double*)it->second[7];
    a.nb = meta[0];
    a.nwin = meta[1];
    a.wins_per_note = meta[2];
    a.num_harmonic = meta[3];
    const int L = meta[4];
    a.out = rows;
    a.argmax = d_argmax;
    a.wbase = (const int*)it->second[9];
    const size_t extra = sizeof(double) * (size_t)(a.nb + a.nwin + 2);
    auto launch = [&](auto kern, int T, size_t lds) -> int {
        if (lds > 160 * 1024) return set_error(ctx, MPX_EUNSUPPORTED, "harmonic energy: frame %d needs %zu B of LDS", N, lds);
        if (lds > 48 * 1024) MPX_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        prof_mark(ctx, stream, "he_blue_kernel");
        hipLaunchKernelGGL(kern, dim3((unsigned)num_frames), dim3(T), lds, stream, a);
        prof_mark(ctx, stream, nullptr);
        MPX_HIP(ctx, hipGetLastError());
        return MPX_OK;
    };
    int rc;
    if (meta[5] > 1) {
        HeBlueSplitArgs sa;
        sa.b = a;
        sa.R = meta[5];
        sa.n1 = meta[6];
        sa.coef = (const cx<double>*)it->second[8];
        const size_t lds = sizeof(cx<double>) * lds_slots(8192) + extra;
        if (lds > 160 * 1024) return set_error(ctx, MPX_EUNSUPPORTED, "harmonic energy: frame %d needs %zu B of LDS", N, lds);
        // two window bins per thread (the shapes the chroma path sends here), eight for the bin lists of very wide windows
        auto kern = a.nb <= 1024 ? he_blue_split_kernel<8192, 512, 2> : he_blue_split_kernel<8192, 512, 8>;
        MPX_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        prof_mark(ctx, stream, "he_blue_kernel");
        hipLaunchKernelGGL(kern, dim3((unsigned)num_frames), dim3(512), lds, stream, sa);
        prof_mark(ctx, stream, nullptr);
        MPX_HIP(ctx, hipGetLastError());
        rc = MPX_OK;
    } else if (L == 1024) rc = launch(he_blue_kernel<1024, 64>, 64, sizeof(cx<double>) * lds_slots(1024) + extra);
    else if (L == 2048) rc = launch(he_blue_kernel<2048, 128>, 128, sizeof(cx<double>) * lds_slots(2048) + extra);
    else if (L == 4096) rc = launch(he_blue_kernel<4096, 256>, 256, sizeof(cx<double>) * lds_slots(4096) + extra);
    else rc = launch(he_blue_kernel<8192, 512>, 512, sizeof(cx<double>) * lds_slots(8192) + extra);
    if (rc) return rc;
    if (d_sum) return segment_sum(ctx, rows, nullptr, 1, num_frames, d_sum, stream);
    return MPX_OK;
}

int he_run(mpx_ctx* ctx, const float* d_signal, int64_t n, const FrameDesc* d_desc, int64_t num_frames,
           int fs, const mpx_he_params* params, int frame, int hop, double* d_chroma_frames,
           double* d_chroma_sum, hipStream_t stream) {
    mpx_he_params p = params ? *params : mpx_he_params{2, 2, 2};
    if (p.num_harmonic < 1 || p.num_octave < 1 || p.num_bins < 0 || p.num_harmonic * p.num_octave > 64)
        return set_error(ctx, MPX_EINVAL, "bad harmonic-energy params (%d,%d,%d)", p.num_harmonic,
                         p.num_octave, p.num_bins);
    if (fs <= 0) return set_error(ctx, MPX_EINVAL, "fs must be positive");
    if (num_frames == 0) return MPX_OK;
    if (frame < 1024 || frame > 16384 || (frame & (frame - 1))) {
        if (frame < 2 || frame > (1 << 18))
            return set_error(ctx, MPX_EUNSUPPORTED, "harmonic energy: frame size %d (supported: 2 ... 262144, as far as 64 passes of the 8192-point chirp-z reach)", frame);
        return he_blue_run(ctx, d_signal, n, d_desc, num_frames, fs, p, frame, hop, d_chroma_frames, d_chroma_sum, stream);
    }
    const bool f32 = ctx->flags & MPX_FLAG_F32;
    auto key = std::make_tuple(fs, frame, p.num_harmonic, p.num_octave, p.num_bins);
    auto it = ctx->he_plans.find(key);
    if (it == ctx->he_plans.end()) {
        HePlan plan;
        int rc = f32 ? he_build_plan<float>(ctx, fs, frame, p, plan) : he_build_plan<double>(ctx, fs, frame, p, plan);
        if (rc) return rc;
        it = ctx->he_plans.emplace(key, plan).first;
    }
    return f32 ? he_dispatch<float>(ctx, it->second, d_signal, n, d_desc, num_frames, frame, hop, d_chroma_frames, d_chroma_sum, stream)
               : he_dispatch<double>(ctx, it->second, d_signal, n, d_desc, num_frames, frame, hop, d_chroma_frames, d_chroma_sum, stream);
}

// Debug tap: the bin of every window's first maximum, [F, nwin] (MultipitchHarmonicEnergy.dft_maxes, harmonic_energy.py:57-65).
// Any frame size runs the chirp-z kernels here (one workgroup per frame): a plot-only attribute, not a tuned path.
int he_argmax_run(mpx_ctx* ctx, const float* d_signal, int64_t n, int64_t num_frames, int fs, const mpx_he_params* params,
                  int frame, int hop, int* d_argmax, int* h_bounds, hipStream_t stream) {
    mpx_he_params p = params ? *params : mpx_he_params{2, 2, 2};
    if (p.num_harmonic < 1 || p.num_octave < 1 || p.num_bins < 0 || p.num_harmonic * p.num_octave > 64)
        return set_error(ctx, MPX_EINVAL, "bad harmonic-energy params (%d,%d,%d)", p.num_harmonic, p.num_octave, p.num_bins);
    if (fs <= 0) return set_error(ctx, MPX_EINVAL, "fs must be positive");
    if (frame < 2 || frame > (1 << 18)) return set_error(ctx, MPX_EUNSUPPORTED, "harmonic energy: frame size %d", frame);
    if (h_bounds) {   // [nwin][2]: k0, k1 as the reference computes them (harmonic_energy.py:50-55)
        HeWindows W;
        if (int rc = he_windows(ctx, fs, frame, p, W)) return rc;
        for (size_t w = 0; w < W.k0.size(); ++w) {
            h_bounds[2 * w] = W.k0[w];
            h_bounds[2 * w + 1] = W.k1[w];
        }
    }
    if (num_frames == 0) return MPX_OK;
    return he_blue_run(ctx, d_signal, n, nullptr, num_frames, fs, p, frame, hop, nullptr, nullptr, stream, d_argmax);
}

}  // namespace mpx
