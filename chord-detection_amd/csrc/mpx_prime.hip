// Prime-multiF0 chroma (reference method 4, prime_multif0.py:41-91) in fp64.
//
// An item is one (candidate frequency, frame): Hann window (numpy.hanning, symmetric), N-point
// DFT of the real frame by Bluestein's chirp-z on the LDS FFT (N = int(8/f*fs) is never a power of
// two), |X|/sum(window) for the lower half of the one-sided spectrum (prime_multif0.py:59-61), then
// harmonic_elim_runs rounds of block argmax -> pitch class -> exact-frequency harmonic elimination
// (prime_multif0.py:66-82).  Every item writes its <= runs (pitch class, value) pairs to a fixed slot;
// one workgroup per clip adds them up in item order (deterministic).
//
// prime_wave_kernel (chirp-z lengths 1024 and 2048, round 5): a WAVE per item, the item in registers -- see the kernel.
// prime_pers_kernel (chirp-z length 4096; 1024 and 2048 in development builds): PERSISTENT workgroups, each bound to one candidate frequency.  What
// depends on the candidate only -- window x chirp, the filter spectrum in the register order of the DIF engine, the
// twiddle bases, the output chirp -- is loaded into registers once; the workgroup then walks that candidate's frames
// across all clips with a stride, the next frame's samples in flight under the current frame's transforms.  The two
// transforms run on the wave-local in-place DIF / inverse-DIT engine (mpx_fft_dif.hpp: one workgroup barrier each, the
// pointwise product in registers with no reordering), the magnitudes never leave registers, and the argmax is a DPP
// maximum + ballot per wave and ONE barrier per round.  (Round 2's kernel, kept for the 8192-point class only, was a
// workgroup per item: seven million workgroups per 4096 clips that each fetched their tables from L2 and ran ~35 barriers.)
#include <cmath>
#include <cstring>

#include "mpx_fft.hpp"
#include "mpx_fft_dif.hpp"
#include "mpx_he_wave.hpp"
#include "mpx_internal.hpp"

namespace mpx {

struct PrimeCand {          // per candidate frequency, device resident
    int N, L, half;         // frame length, Bluestein FFT length, bins kept = int((N//2+1)/2)
    int paired;             // prime_pers_kernel: two frames per transform (the filter then covers the outputs -(half-1) .. half-1)
    double val;             // frequency step: 1.0/(N*(1/fs)) exactly as numpy.fft.fftfreq builds it
    double wsum;            // sum(numpy.hanning(N))
    const double* win;      // [N]
    const cx<double>* chirp;  // [N]
    const cx<double>* bhat;   // [L] FFT_L(chirp filter)/L, natural order (8192-point class)
    const cx<double>* bhat_r; // [8][L/8] the same in the register order dif_fft_keep_last<L> leaves (L <= 4096)
    const cx<double>* tw;     // [L] W_L
    int R, n1;                // prime_kernel, frames whose chirp-z does not fit 8192 points: input decimated by R, n1 = ceil(N / R)
    const cx<double>* coef;   // [R][half] W_N^(r k) conj(chirp[k]) (R > 1; chirp is then exp(i pi R j^2 / N), j < max(n1, half))
    // prime_wave_kernel (L <= 2048), WD = L / 32 lanes per transform:
    const cx<double>* wv_wc;  // [22][WD] window x conj(chirp) at n = lane + WD n1 (0 from N on)
    const cx<double>* wv_fr;  // [32][WD] bhat where the wave's forward transform leaves it: register p, lane
    const cx<double>* wv_oc;  // [6][WD]  conj(chirp[k])^2, k = lane + WD q (0 from `half` on)
    const int* wv_pc;         // [6 WD]   pitch class of bin k (before the note-name quirk), -2: its frequency has none (bin 0)
    const cx<double>* wv_theta;   // [5][32] W_1024^(k1 (16 >> s)): the stage constants of the modulated transform of row k1
    const cx<double>* wv_tw2;     // [32][33] W_2048^(k1 + 32 br5(p)) at [p][k1], 1 at [p][32] (L = 2048)
};

struct PrimeItem {
    long long start;        // first sample of the frame in the packed signal
    int valid;              // samples that exist (rest is frame_cutter's zero padding)
    int cand;
    long long slot;         // output slot (clip-major, reference loop order)
    int valid_b, pad;       // prime_pers_kernel: the NEXT frame of the same clip and candidate rides along (0: there is none);
                            // it starts N samples later and its slot is slot + 1
};

constexpr int PRIME_MAX_RUNS = 4;

template <int L, int T>
__global__ __launch_bounds__(T) void prime_kernel(const float* __restrict__ sig, const PrimeItem* __restrict__ items,
                                                  const PrimeCand* __restrict__ cands, int runs, int elim,
                                                  int note_names, int* out_pc, double* out_val, int per_clip, long long clip_len,
                                                  long long clip_slots) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cx<double>* buf = reinterpret_cast<cx<double>*>(smem);
    // the magnitudes (half <= N/2 + 1 <= L/4 + 1 doubles) go into the upper half of the FFT buffer: they are computed
    // from slots below 4.25 L bytes, so nothing they overwrite is still needed -- and the 8192-point class fits
    double* mag = reinterpret_cast<double*>(smem + 8 * (size_t)L);
    __shared__ double red_v[T];
    __shared__ int red_i[T];
    const int tid = threadIdx.x;
    // per_clip > 0: every clip has the same length, `items` holds ONE clip's items of this class and workgroup b is
    // item b % per_clip of clip b / per_clip (the host then builds and uploads 1/clips of the list)
    PrimeItem it = items[per_clip ? blockIdx.x % per_clip : blockIdx.x];
    if (per_clip) {
        const long long clip = blockIdx.x / per_clip;
        it.start += clip * clip_len;
        it.slot += clip * clip_slots;
    }
    const PrimeCand c = cands[it.cand];
    const int N = c.N, half = c.half;
    const float* __restrict__ x = sig + it.start;
    cx<double> regs[L / T];

    if (c.R <= 1) {
    for (int n = tid; n < L; n += T) {
        cx<double> v = {0.0, 0.0};
        if (n < N) {
            const double s = (n < it.valid ? (double)x[n] : 0.0) * c.win[n];
            const cx<double> ch = c.chirp[n];
            v = {s * ch.x, -s * ch.y};  // s * conj(chirp)
        }
        buf[lds_slot(n)] = v;
    }
    __syncthreads();
    fft_lds<L, T, false, double>(buf, c.tw, regs, tid);
    for (int k = tid; k < L; k += T) {
        const cx<double> p = cmul(buf[lds_slot(k)], c.bhat[k]);
        buf[lds_slot(k)] = {p.y, p.x};  // swapped: the next forward FFT acts as the inverse
    }
    __syncthreads();
    fft_lds<L, T, false, double>(buf, c.tw, regs, tid);
    for (int k = tid; k < half; k += T) {
        const cx<double> b = buf[lds_slot(k)];
        const cx<double> ch = c.chirp[k];
        const cx<double> z = cmul(cx<double>{b.y, b.x}, cx<double>{ch.x, -ch.y});
        mag[k] = hypot(z.x, z.y) / c.wsum;  // mlab: np.abs(result) / window.sum()
    }
    __syncthreads();
    } else {
        // Frames above 6553 samples (input above ~107 kHz): N + half - 1 points do not fit the engine.  The input decimated by
        // R: X[k] = sum_r W_N^(r k) G_r[k], G_r[k] = sum_m s[R m + r] W_N^(R m k) a chirp-z of n1 = ceil(N / R) points with
        // the chirp exp(i pi R j^2 / N); R passes, a thread accumulates its bins k = tid + j T < half <= 4096 over them.
        constexpr int PB = 4096 / T;
        cx<double> acc[PB];
#pragma unroll
        for (int j = 0; j < PB; ++j) acc[j] = {0.0, 0.0};
        for (int r = 0; r < c.R; ++r) {
            for (int m = tid; m < L; m += T) {
                cx<double> v = {0.0, 0.0};
                const int n = c.R * m + r;
                if (m < c.n1 && n < N) {
                    const double s = (n < it.valid ? (double)x[n] : 0.0) * c.win[n];
                    const cx<double> ch = c.chirp[m];
                    v = {s * ch.x, -s * ch.y};
                }
                buf[lds_slot(m)] = v;
            }
            __syncthreads();
            fft_lds<L, T, false, double>(buf, c.tw, regs, tid);
            for (int k = tid; k < L; k += T) {
                const cx<double> p = cmul(buf[lds_slot(k)], c.bhat[k]);
                buf[lds_slot(k)] = {p.y, p.x};
            }
            __syncthreads();
            fft_lds<L, T, false, double>(buf, c.tw, regs, tid);
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                const int k = tid + j * T;
                if (k < half) {
                    const cx<double> b = buf[lds_slot(k)];
                    const cx<double> z = cmul(cx<double>{b.y, b.x}, c.coef[(size_t)r * half + k]);
                    acc[j] = {acc[j].x + z.x, acc[j].y + z.y};
                }
            }
            __syncthreads();   // the next pass (and the magnitudes) overwrite buf
        }
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const int k = tid + j * T;
            if (k < half) mag[k] = hypot(acc[j].x, acc[j].y) / c.wsum;
        }
        __syncthreads();
    }

    for (int run = 0; run < runs; ++run) {
        // numpy argmax: first index of the maximum
        double bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int k = tid; k < half; k += T) {
            const double v = mag[k];
            if (v > bv) {
                bv = v;
                bi = k;
            }
        }
        red_v[tid] = bv;
        red_i[tid] = bi;
        __syncthreads();
        for (int s = T / 2; s > 0; s >>= 1) {
            if (tid < s) {
                const double ov = red_v[tid + s];
                const int oi = red_i[tid + s];
                if (ov > red_v[tid] || (ov == red_v[tid] && oi < red_i[tid])) {
                    red_v[tid] = ov;
                    red_i[tid] = oi;
                }
            }
            __syncthreads();
        }
        if (tid == 0) {
            int pc = -1;
            double val = 0.0;
            if (half > 0) {
                const int idx = red_i[0] == 0x7fffffff ? 0 : red_i[0];
                const double max_f = (double)idx * c.val;
                const double midi = 12.0 * (log2(max_f) - log2(440.0)) + 69.0;
                // hz_to_note raises on NaN (ValueError) and on +-inf (OverflowError, e.g. the DC bin): the
                // reference `continue`s: nothing is added and nothing is eliminated (prime_multif0.py:73-74)
                if (midi == midi && !isinf(midi)) {
                    const long long note = (long long)nearbyint(midi);
                    pc = (int)(((note % 12) + 12) % 12);
                    val = mag[idx];
                    for (int k = 1; k < elim; ++k) {
                        const double target = (double)k * max_f;  // f == k * max_f, exact comparison (:80)
                        for (int j = k * idx - 1; j <= k * idx + 1; ++j)
                            if (j >= 0 && j < half && (double)j * c.val == target) mag[j] = 0.0;
                    }
                    // unicode-sharp quirk A.18 (MPX_NOTES_UNICODE): sharps land in a stray key and are lost, but the
                    // elimination above has happened; ASCII note names (librosa < 0.8) keep every pitch class
                    if (note_names == MPX_NOTES_UNICODE && (pc == 1 || pc == 3 || pc == 6 || pc == 8 || pc == 10)) pc = -1;
                }
            }
            out_pc[it.slot * PRIME_MAX_RUNS + run] = pc;
            out_val[it.slot * PRIME_MAX_RUNS + run] = val;
        }
        __syncthreads();
    }
}


// ---- persistent kernel -------------------------------------------------------------------------------------------
struct PrimeWork {          // per workgroup
    int cand;               // candidate frequency this workgroup is bound to
    int worker, workers;    // it takes items worker, worker + workers, ... of that candidate
    int item0, count;       // the candidate's items in `items`: [item0, item0 + count) (of ONE clip when the clips are uniform)
};

template <int CTRL>
__device__ __forceinline__ double prime_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double prime_readlane(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// maximum over the wave of values that are never NaN (every lane gets it)
__device__ __forceinline__ double prime_wave_max(double v) {
    v = fmax(v, prime_dpp<0xB1>(v));    // quad_perm [1,0,3,2]
    v = fmax(v, prime_dpp<0x4E>(v));    // quad_perm [2,3,0,1]
    v = fmax(v, prime_dpp<0x141>(v));   // row_half_mirror
    v = fmax(v, prime_dpp<0x140>(v));   // row_mirror: all 16 lanes of a row agree
    return fmax(fmax(prime_readlane(v, 0), prime_readlane(v, 16)), fmax(prime_readlane(v, 32), prime_readlane(v, 48)));
}

template <int L>
__global__ __launch_bounds__(L / 8) void prime_pers_kernel(const float* __restrict__ sig, const PrimeItem* __restrict__ items,
                                                           const PrimeWork* __restrict__ work, const PrimeCand* __restrict__ cands,
                                                           int runs, int elim, int note_names, int* out_pc, double* out_val,
                                                           int uniform_clips, long long clip_len, long long clip_slots) {
    constexpr int T = L / 8, NW = T / 64;
    static_assert(T % 64 == 0, "whole waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cx<double>* buf = reinterpret_cast<cx<double>*>(smem);
    __shared__ cx<double> exch[2 * T];   // y[6T .. 8T): the negative output indices of the chirp-z, for the split below
    __shared__ double red_v[2][2][NW];
    __shared__ int red_i[2][2][NW];
    __shared__ int nonzero[2][2];        // [parity][frame]: some sample of the frame is not zero
    const int tid = threadIdx.x;
    const PrimeWork wk = work[blockIdx.x];
    const PrimeCand c = cands[wk.cand];
    const int N = c.N, half = c.half;
    if (tid < 4) nonzero[tid >> 1][tid & 1] = 0;
    // what depends on the candidate only, once per workgroup
    cx<double> wc[8], bh[8], oc[2];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int n = tid + r * T;
        wc[r] = {0.0, 0.0};
        if (n < N) {
            const double w = c.win[n];
            const cx<double> ch = c.chirp[n];
            wc[r] = {w * ch.x, -(w * ch.y)};   // window x conj(chirp)
        }
        bh[r] = c.bhat_r[r * T + tid];
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int k = tid + r * T;
        oc[r] = {0.0, 0.0};
        if (k < half) {
            const cx<double> ch = c.chirp[k];
            oc[r] = {ch.x, -ch.y};
        }
    }
    const DifTwiddles<L, double> twd = dif_load_twiddles<L, double>(c.tw, tid);
    const long long total = uniform_clips ? (long long)wk.count * uniform_clips : wk.count;
    __syncthreads();

    auto item_at = [&](long long i) -> PrimeItem {
        PrimeItem it;
        if (uniform_clips) {
            const long long clip = i / wk.count;
            it = items[wk.item0 + (int)(i - clip * wk.count)];
            it.start += clip * clip_len;
            it.slot += clip * clip_slots;
        } else {
            it = items[wk.item0 + i];
        }
        return it;
    };
    // the two frames of an item follow each other in the clip: frame b starts N samples after frame a
    auto fetch = [&](const PrimeItem& it, float* xa, float* xb) {
        const float* __restrict__ x = sig + it.start;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int n = tid + r * T;
            xa[r] = n < it.valid ? x[n] : 0.0f;
            xb[r] = n < it.valid_b ? x[N + n] : 0.0f;
        }
    };

    long long i = wk.worker;
    if (i >= total) return;
    PrimeItem it = item_at(i);
    float xa[8], xb[8];
    fetch(it, xa, xb);
    int par = 0, fpar = 0;   // parity of the argmax slots (per round) and of the zero-frame flags (per item)
    for (; i < total; i += wk.workers) {
        int t = tid;
        asm volatile("" : "+v"(t));   // addresses are rebuilt per item: hoisted, they cost more registers than they save
        // TWO real frames per transform: u = (a + i b) x window x conj(chirp); their spectra are separated afterwards by
        // the conjugate symmetry of a real frame's spectrum, X_a[k] = (X[k] + conj X[-k]) / 2, X_b[k] = (X[k] - conj X[-k]) / 2i
        cx<double> regs[8];
        bool nza = false, nzb = false;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const double a = (double)xa[r], b = (double)xb[r];
            nza |= xa[r] != 0.0f;
            nzb |= xb[r] != 0.0f;
            regs[r] = {a * wc[r].x - b * wc[r].y, a * wc[r].y + b * wc[r].x};
        }
        // An all-zero frame has an all-zero spectrum in the reference; next to a loud partner it would come out as that
        // partner's rounding noise (1e-17), so such a frame is flagged and its magnitudes are set to the exact zeros.
        if (nza) nonzero[fpar][0] = 1;
        if (nzb) nonzero[fpar][1] = 1;
        // the next item's samples travel under this item's transforms
        const long long inext = i + wk.workers;
        const PrimeItem itn = item_at(inext < total ? inext : i);
        fetch(itn, xa, xb);
        dif_fft_keep_last<L, double>(buf, twd, regs, t);     // (one workgroup barrier inside: the flags are visible after it)
        const bool live_a = nonzero[fpar][0] != 0, live_b = nonzero[fpar][1] != 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) regs[e] = cmul(regs[e], bh[e]);
        idit_fft_from_last<L, double>(buf, twd, regs, t);     // regs[r] = y[tid + r T]
        if (tid < 2) nonzero[fpar][tid] = 0;                  // everybody has read them (barrier inside the inverse transform)
        fpar ^= 1;
        exch[t] = regs[6];
        exch[t + T] = regs[7];
        __syncthreads();
        // y[-k] for k = tid (r = 0) and k = tid + T (r = 1): positions L - k
        const cx<double> ym0 = t == 0 ? regs[0] : exch[2 * T - t];
        const cx<double> ym1 = t == 0 ? exch[T] : exch[T - t];
        // X[k] = conj(chirp[k]) y[k], X[-k] = conj(chirp[k]) y[-k] (the chirp is even); mlab: np.abs(result) / window.sum()
        double ma[2], mb[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const cx<double> xp = cmul(regs[r], oc[r]), xm = cmul(r == 0 ? ym0 : ym1, oc[r]);
            // (a candidate whose 1.5 N does not fit 4096 points runs one frame per transform: b = 0 and X_a = X[k] itself;
            //  its filter does not cover the negative outputs, xm is then meaningless and unused)
            const cx<double> sa = c.paired ? cx<double>{0.5 * (xp.x + xm.x), 0.5 * (xp.y - xm.y)} : xp;    // X_a = (X[k] + conj X[-k]) / 2
            const cx<double> sb = {0.5 * (xp.y + xm.y), 0.5 * (xm.x - xp.x)};    // X_b = (X[k] - conj X[-k]) / 2i
            ma[r] = live_a ? hypot(sa.x, sa.y) / c.wsum : 0.0;
            mb[r] = live_b ? hypot(sb.x, sb.y) / c.wsum : 0.0;
        }
        const bool has1 = tid + T < half, has0 = tid < half;
        const bool two = it.valid_b > 0;
        for (int run = 0; run < runs; ++run) {
            // numpy argmax: first index of the maximum (NaN never wins, as before); both frames through the same barrier
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const double* m = f ? mb : ma;
                double bv = -INFINITY;
                if (has0 && m[0] > bv) bv = m[0];
                if (has1 && m[1] > bv) bv = m[1];
                const double wmax = prime_wave_max(bv);
                const unsigned long long b0 = __ballot(has0 && m[0] == wmax), b1 = __ballot(has1 && m[1] == wmax);
                const int wbase = tid & ~63;
                int widx = 0x7fffffff;
                if (wmax > -INFINITY) widx = b0 ? wbase + __builtin_ctzll(b0) : wbase + T + __builtin_ctzll(b1);
                if ((tid & 63) == 0) {
                    red_v[par][f][tid >> 6] = wmax;
                    red_i[par][f][tid >> 6] = widx;
                }
            }
            __syncthreads();
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                double* m = f ? mb : ma;
                double best = red_v[par][f][0];
                int idx = red_i[par][f][0];
#pragma unroll
                for (int w = 1; w < NW; ++w) {
                    const double ov = red_v[par][f][w];
                    const int oi = red_i[par][f][w];
                    if (ov > best || (ov == best && oi < idx)) {
                        best = ov;
                        idx = oi;
                    }
                }
                int pc = -1;
                double val = 0.0;
                if (half > 0) {
                    const bool none = idx == 0x7fffffff;   // nothing compared greater than -inf: every magnitude is NaN
                    if (none) idx = 0;
                    const double max_f = (double)idx * c.val;
                    const double midi = 12.0 * (log2(max_f) - log2(440.0)) + 69.0;
                    // hz_to_note raises on NaN (ValueError) and on +-inf (OverflowError, e.g. the DC bin): the
                    // reference `continue`s: nothing is added and nothing is eliminated (prime_multif0.py:73-74)
                    if (midi == midi && !isinf(midi)) {
                        const long long note = (long long)nearbyint(midi);
                        pc = (int)(((note % 12) + 12) % 12);
                        val = none ? __builtin_nan("") : best;
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            const int j = tid + r * T;
                            for (int k = 1; k < elim; ++k) {   // f == k * max_f, exact comparison (:80), among bins k idx - 1 .. k idx + 1
                                const int d = j - k * idx;
                                if (j < half && d >= -1 && d <= 1 && (double)j * c.val == (double)k * max_f) m[r] = 0.0;
                            }
                        }
                        // unicode-sharp quirk A.18 (MPX_NOTES_UNICODE): sharps land in a stray key and are lost, but the
                        // elimination above has happened; ASCII note names (librosa < 0.8) keep every pitch class
                        if (note_names == MPX_NOTES_UNICODE && (pc == 1 || pc == 3 || pc == 6 || pc == 8 || pc == 10)) pc = -1;
                    }
                }
                if (tid == 0 && (f == 0 || two)) {
                    out_pc[(it.slot + f) * PRIME_MAX_RUNS + run] = pc;
                    out_val[(it.slot + f) * PRIME_MAX_RUNS + run] = val;
                }
            }
            par ^= 1;
        }
        it = itn;
    }
}

// ---- wave kernel ---------------------------------------------------------------------------------------------------
// prime_wave_kernel<L> (round 5; chirp-z lengths 1024 and 2048: every candidate of 22.05 kHz input, the upper octave of
// 44.1 kHz): the structure of he_wave_kernel.  A chirp-z is a forward and an inverse transform of L points; here both are
// 1024-point transforms of one LANE CLASS (the 32 lanes of one parity, 32 complex points per lane in registers): a 32-point
// DFT in registers, one transpose through LDS inside the class, a second, modulated 32-point DFT (mpx_he_wave.hpp).  In:
// lane l, register n1 = point l + 32 n1; out: lane k1, register p = frequency k1 + 32 br5(p) -- which, read as "lane,
// br5(register)", is the input layout again: the inverse transform (the forward one on swapped components) starts from a
// renaming of registers, the filter spectrum is multiplied on where it stands (table in register order), and nothing but
// the two transposes of each transform touches LDS.
//   L = 1024: the two lane classes of a wave run two ITEMS (pairs of frames of the same candidate) side by side;
//   L = 2048: one item on both classes -- even lanes transform u[2m], odd lanes u[2m + 1]; the radix-2 step that joins them
//             (U[k], U[k + 1024] = E[k] +- W_2048^k O[k]) and its mirror image in front of the inverse transform pair lanes
//             2 k1 and 2 k1 + 1 on the same register: quad-permute DPP, no LDS.
// A wave never waits for another wave: no workgroup barrier in the loop (prime_pers_kernel: seven per item), the arg-max
// is DPP + ballot inside the wave.  Per lane six output bins k = lane + WD q (half <= 6 WD) and their mirrors -k, which the
// inverse transform leaves in registers 26..31 of the mirrored lane (one 6-register exchange through the wave's buffer).
// The arg-max runs on |X|^2 (monotone in the magnitude the reference compares, prime_multif0.py:68); the square root is
// taken of the winner only.
constexpr int PW_NR = 22;   // registers that can hold samples: 22 WD >= the longest frame of a class (684 / 1366)
constexpr int PW_NQ = 6;    // output bins per lane: 6 WD >= half
// Waves per workgroup = per CU: what 160 KB of LDS hold next to the tables (33 / 80 KB).  Measured (4096 clips, profiles/r5/prime_wave_ab.txt):
// 1024 points: 4 / 6 / 7 waves 16.1 / 16.0 / 13.9 ms; 2048 points: 2 / 3 / 4 waves 31.6 / 21.2 / 16.0 ms (a wave per SIMD scales, the
// kernel is then bound by one wave's issue rate); a fifth wave -- output chirp from global memory, every wave back under 256
// registers -- 19.7 ms.
constexpr int PW_WAVES_1024 = 7, PW_WAVES_2048 = 4;
template <int L>
__host__ __device__ constexpr int pw_table_elems() {
    return (PW_NR + 32 + PW_NQ) * (L / 32) + 160 + (L == 2048 ? 32 * 33 : 0) + PW_NQ * (L / 32) / 4;   // (the last: PW_NQ WD ints)
}

template <int CTRL>
__device__ __forceinline__ double pw_dpp(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// maximum over the lanes of the caller's class (L = 1024: its parity; L = 2048: the wave) of values that are never NaN
template <int L>
__device__ __forceinline__ double pw_class_max(double v, int parity) {
    v = fmax(v, pw_dpp<0x4E>(v));                        // quad_perm [2,3,0,1]: lane ^ 2
    if constexpr (L == 2048) v = fmax(v, pw_dpp<0xB1>(v));   // quad_perm [1,0,3,2]: lane ^ 1
    v = fmax(v, pw_dpp<0x124>(v));                       // row_ror:4
    v = fmax(v, pw_dpp<0x128>(v));                       // row_ror:8: the lanes of a row that agree modulo 4 (2) agree
    const double e = fmax(fmax(prime_readlane(v, 0), prime_readlane(v, 16)), fmax(prime_readlane(v, 32), prime_readlane(v, 48)));
    if constexpr (L == 2048) return e;
    const double o = fmax(fmax(prime_readlane(v, 1), prime_readlane(v, 17)), fmax(prime_readlane(v, 33), prime_readlane(v, 49)));
    return parity ? o : e;
}

template <int CTRL>
__device__ __forceinline__ int pw_dpp_i(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true); }
template <int L>
__device__ __forceinline__ int pw_class_min(int v, int parity) {
    v = min(v, pw_dpp_i<0x4E>(v));
    if constexpr (L == 2048) v = min(v, pw_dpp_i<0xB1>(v));
    v = min(v, pw_dpp_i<0x124>(v));
    v = min(v, pw_dpp_i<0x128>(v));
    const int e = min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
    if constexpr (L == 2048) return e;
    const int o = min(min(__builtin_amdgcn_readlane(v, 1), __builtin_amdgcn_readlane(v, 17)), min(__builtin_amdgcn_readlane(v, 33), __builtin_amdgcn_readlane(v, 49)));
    return parity ? o : e;
}

// one transform of 1024 points per lane class, in place: z[n1] = point l + 32 n1 (l = lane >> 1) in, z[p] = frequency
// (lane >> 1) + 32 br5(p) out
template <bool FIRST_LEVEL_DONE = false>
__device__ __forceinline__ void pw_fft1024(cx<double>* z, char* xbuf, const cx<double>* theta_lds) {
    if constexpr (FIRST_LEVEL_DONE) hw_fft32_ct_after_first_level(z);
    else hw_fft32_ct(z);
    hw_phase();
    cx<double> th[5];
    const int ol = hw_opaque((int)(threadIdx.x & 63));
    char* wr = xbuf + 8 * ol;
    const char* rd = xbuf + HW_PAIR * (ol >> 1) + 8 * (ol & 1);
#pragma unroll
    for (int st = 0; st < 5; ++st) th[st] = theta_lds[32 * st + (ol >> 1)];
    cx<double> b[32];
#pragma unroll
    for (int p = 0; p < 32; ++p) *reinterpret_cast<double*>(wr + HW_PAIR * hw_br5(p)) = z[p].x;
    wave_lds_fence();
#pragma unroll
    for (int c = 0; c < 32; ++c) b[c].x = *reinterpret_cast<const double*>(rd + 16 * c);
    wave_lds_fence();
#pragma unroll
    for (int p = 0; p < 32; ++p) *reinterpret_cast<double*>(wr + HW_PAIR * hw_br5(p)) = z[p].y;
    wave_lds_fence();
#pragma unroll
    for (int c = 0; c < 32; ++c) b[c].y = *reinterpret_cast<const double*>(rd + 16 * c);
    wave_lds_fence();
    hw_phase();
    hw_fft32_modulated(b, th);
    hw_phase();
#pragma unroll
    for (int p = 0; p < 32; ++p) z[p] = b[p];
}

template <int L, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void prime_wave_kernel(const float* __restrict__ sig, const PrimeItem* __restrict__ items,
                                                                    const PrimeWork* __restrict__ work, const PrimeCand* __restrict__ cands,
                                                                    int runs, int elim, int note_names, int* out_pc, double* out_val,
                                                                    int uniform_clips, long long clip_len, long long clip_slots) {
    constexpr int WD = L / 32, T = WAVES * 64, TAB = pw_table_elems<L>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cx<double>* wc_lds = reinterpret_cast<cx<double>*>(smem);   // [PW_NR][WD]
    cx<double>* fr_lds = wc_lds + PW_NR * WD;                   // [32][WD]
    cx<double>* oc_lds = fr_lds + 32 * WD;                      // [PW_NQ][WD] conj(chirp[k])^2
    cx<double>* theta_lds = oc_lds + PW_NQ * WD;                // [5][32]
    [[maybe_unused]] cx<double>* tw2_lds = theta_lds + 160;     // [32][33] (L = 2048)
    int* pc_lds = reinterpret_cast<int*>(tw2_lds + (L == 2048 ? 32 * 33 : 0));   // [PW_NQ WD] pitch class of bin k, -2: none
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    char* xbuf = smem + (size_t)TAB * 16 + wave * HW_XBUF;
    const PrimeWork wk = work[blockIdx.x];
    const PrimeCand c = cands[wk.cand];
    const int N = c.N, half = c.half;
    for (int i = tid; i < PW_NR * WD; i += T) wc_lds[i] = c.wv_wc[i];
    for (int i = tid; i < 32 * WD; i += T) fr_lds[i] = c.wv_fr[i];
    for (int i = tid; i < PW_NQ * WD; i += T) oc_lds[i] = c.wv_oc[i];
    for (int i = tid; i < PW_NQ * WD; i += T) pc_lds[i] = c.wv_pc[i];
    for (int i = tid; i < 160; i += T) theta_lds[i] = c.wv_theta[i];
    if constexpr (L == 2048)
        for (int i = tid; i < 32 * 33; i += T) tw2_lds[i] = c.wv_tw2[i];
    __syncthreads();   // the only one

    const int parity = lane & 1;
    const int lw = L == 1024 ? lane >> 1 : lane;   // this lane's index inside its transform
    const double scale = 0.5 / c.wsum;
    constexpr int PER = L == 1024 ? 2 : 1;         // items per wave and iteration
    const int count = wk.count;
    const int worker = wk.worker + __builtin_amdgcn_readfirstlane(wave);
    const int stride = wk.workers * PER;           // (the host keeps count + stride below 2^31)
    // Position of this lane's class: item i of the candidate's list, or -- equal-length clips -- item j of clip `clip`,
    // which follows from the first clip's first item (frames 2j and 2j + 1 of the clip)
    int i = worker * PER + (L == 1024 ? parity : 0), clip = 0;
    int dclip = 0, dj = 0;
    long long start0 = 0, slot0 = 0;
    if (uniform_clips) {
        clip = i / count;
        i -= clip * count;
        dclip = stride / count;
        dj = stride - dclip * count;
        start0 = items[wk.item0].start;
        slot0 = items[wk.item0].slot;
    }
    struct Cur {
        const float* xa;
        int va, vb;
        long long slot;
    };
    auto live_here = [&]() -> bool { return uniform_clips ? clip < uniform_clips : i < count; };
    auto item_here = [&]() -> Cur {
        Cur it{sig, 0, 0, -1};
        if (live_here()) {
            long long start;
            if (uniform_clips) {
                const long long s = (long long)i * 2 * N, left = clip_len - s, lb = left - N;
                it.va = (int)(left >= N ? N : left);
                it.vb = lb > 0 ? (int)(lb >= N ? N : lb) : 0;
                start = start0 + clip * clip_len + s;
                it.slot = slot0 + clip * clip_slots + 2 * i;
            } else {
                const PrimeItem p = items[wk.item0 + i];
                start = p.start;
                it.va = p.valid;
                it.vb = p.valid_b;
                it.slot = p.slot;
            }
            it.xa = sig + start;
        }
        return it;
    };
    auto advance = [&]() {
        i += stride;
        if (uniform_clips) {
            i -= stride - dj;
            clip += dclip;
            if (i >= count) {
                i -= count;
                ++clip;
            }
        }
    };
    // A frame's samples through a buffer descriptor of exactly the samples that exist (base = the frame, size = 4 x valid):
    // positions past them -- the zero padding of a last frame, the registers past N, a frame b that is not there, an item
    // past the end of the list -- come back as zeros from the hardware's range check: no clamp, no select, no branch.
    // Frame b starts N samples after frame a.  The two lane classes of the 1024-point kernel hold different items: one
    // descriptor pair per class, each class's loads under its own half of EXEC into the same registers.
    auto fetch_class = [&](int src_lane, float* xa, float* xb, const Cur& it) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)(uintptr_t)it.xa & 0xffffffffull), src_lane);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)(uintptr_t)it.xa >> 32), src_lane);
        const int va = __builtin_amdgcn_readlane(it.va, src_lane), vb = __builtin_amdgcn_readlane(it.vb, src_lane);
        const float* pa = reinterpret_cast<const float*>((uintptr_t)(((unsigned long long)hi << 32) | lo));
        const auto ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pa), 0, 4 * va, 0x00020000);
        const auto rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pa + (vb > 0 ? N : 0)), 0, 4 * vb, 0x00020000);
        // (the register's part of the position travels in the instruction's offset field or the vector offset: the range
        //  check does not see a scalar offset)
        const int voff = 4 * hw_opaque(lw);
#pragma unroll
        for (int n1 = 0; n1 < PW_NR; ++n1) {
            xa[n1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, voff + 4 * WD * n1, 0, 0));
            xb[n1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, voff + 4 * WD * n1, 0, 0));
        }
    };
    auto fetch = [&](const Cur& it, float* xa, float* xb) {
        if constexpr (L == 1024) {   // (two ifs, not if / else: the compiler merges the arms of an if / else into one load with a per-lane descriptor and serialises it)
            if (parity == 0) fetch_class(0, xa, xb, it);
            hw_phase();
            if (parity == 1) fetch_class(1, xa, xb, it);
        } else {
            fetch_class(0, xa, xb, it);
        }
    };

    if (__builtin_amdgcn_readfirstlane((int)!live_here())) return;   // (lane 0: the even class; wave-uniform)
    Cur it = item_here();
    float xa[PW_NR], xb[PW_NR];
    fetch(it, xa, xb);
    for (;;) {
        cx<double> z[32];
        bool nza = false, nzb = false;
        {
            const int ol = hw_opaque(lw);
#pragma unroll
            for (int n1 = 0; n1 < PW_NR; ++n1) {
                const cx<double> w = wc_lds[n1 * WD + ol];
                const double a = (double)xa[n1], b = (double)xb[n1];
                nza |= xa[n1] != 0.0f;
                nzb |= xb[n1] != 0.0f;
                z[n1] = {a * w.x - b * w.y, a * w.y + b * w.x};   // (a + i b) x window x conj(chirp)
                if ((n1 & 7) == 7) hw_phase();
            }
            // registers 22 .. 31 are zeros: the transform's first level (points n1 and n1 + 16, twiddle 1) copies where it would add
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                if (n1 + 16 < PW_NR) {
                    const cx<double> u = z[n1], v = z[n1 + 16];
                    z[n1] = cadd(u, v);
                    z[n1 + 16] = csub(u, v);
                } else {
                    z[n1 + 16] = z[n1];
                }
            }
        }
        // An all-zero frame has an all-zero spectrum in the reference; next to a loud partner it would come out as that
        // partner's rounding noise, so such a frame is flagged and its magnitudes are the exact zeros (as in prime_pers_kernel).
        // (the class's half of the ballot by SCALAR masks and a select on the parity: as a per-lane 64-bit mask the constant sat in
        //  two register pairs of a kernel that has none to spare, was spilled, and was reloaded behind the prefetch)
        const unsigned long long bal_a = __ballot(nza), bal_b = __ballot(nzb);
        bool live_a, live_b;
        if constexpr (L == 1024) {
            // (readfirstlane: the four answers are pinned as scalars -- left alone the compiler folds the select back into the mask)
            const int ea = __builtin_amdgcn_readfirstlane((int)((bal_a & 0x5555555555555555ull) != 0));
            const int oa = __builtin_amdgcn_readfirstlane((int)((bal_a & 0xAAAAAAAAAAAAAAAAull) != 0));
            const int eb = __builtin_amdgcn_readfirstlane((int)((bal_b & 0x5555555555555555ull) != 0));
            const int ob = __builtin_amdgcn_readfirstlane((int)((bal_b & 0xAAAAAAAAAAAAAAAAull) != 0));
            live_a = (parity ? oa : ea) != 0;
            live_b = (parity ? ob : eb) != 0;
        } else {
            live_a = bal_a != 0;
            live_b = bal_b != 0;
        }
        const long long cur_slot = it.slot;
        const bool cur_b = it.vb > 0;
        hw_phase();
        pw_fft1024<true>(z, xbuf, theta_lds);   // z[p] = U_class[(lane >> 1) + 32 br5(p)]
        hw_phase();
        {
            const int ol = hw_opaque(lw);
            // L = 2048: the twiddle of this lane's row -- W_2048^k in the odd lanes, 1 in the even ones (entry 32 of every
            // table row), so that both kinds run the same instructions: no select on data
            int twi = 0;
            double sgn = 1.0;
            if constexpr (L == 2048) {
                const int ol2 = hw_opaque(lane);
                twi = (ol2 & 1) ? ol2 >> 1 : 32;
                sgn = (ol2 & 1) ? -1.0 : 1.0;
            }
#pragma unroll
            for (int p = 0; p < 32; ++p) {
                const cx<double> f = fr_lds[p * WD + ol];
                if constexpr (L == 1024) {
                    z[p] = cmul(z[p], f);
                } else {
                    // even lane: E, odd lane: O of the same frequency.  s = E | W^k O;  U = E +- W^k O;  V = U x filter;
                    // then the mirror image for the inverse transform: G_even = V_lo + V_hi, G_odd = (V_lo - V_hi) conj(W^k)
                    const cx<double> tw = tw2_lds[p * 33 + twi];
                    const cx<double> s = cmul(z[p], tw);
                    const cx<double> o = {hw_quad_xor<1>(s.x), hw_quad_xor<1>(s.y)};
                    const cx<double> u = {fma(sgn, s.x, o.x), fma(sgn, s.y, o.y)};        // E + t | E - t
                    const cx<double> v = cmul(u, f);
                    const cx<double> ov = {hw_quad_xor<1>(v.x), hw_quad_xor<1>(v.y)};
                    const cx<double> g = {fma(sgn, v.x, ov.x), fma(sgn, v.y, ov.y)};      // V_lo + V_hi | V_lo - V_hi
                    z[p] = {fma(g.x, tw.x, g.y * tw.y), fma(g.y, tw.x, -(g.x * tw.y))};  // g conj(tw)
                }
                if ((p & 7) == 7) hw_phase();
            }
        }
        // inverse transform = the forward one on swapped components; register p holds k2 = br5(p): a renaming
        cx<double> w[32];
#pragma unroll
        for (int n1 = 0; n1 < 32; ++n1) w[n1] = {z[hw_br5(n1)].y, z[hw_br5(n1)].x};
        hw_phase();
        pw_fft1024(w, xbuf, theta_lds);   // (swapped) w[p] = y[lw + WD br5(p)]
        hw_phase();
        // the mirrors: y[-k] for k = lw + WD q sits in register br5(31 - q) of lane WD - lw (lane 0: register br5(32 - q) of its own)
        cx<double> yp[PW_NQ], ym[PW_NQ];
        {
            const int ol = hw_opaque(lane);
            cx<double>* ex = reinterpret_cast<cx<double>*>(xbuf);   // [7][64]; row 6 is never written and never used
#pragma unroll
            for (int r = 0; r < 6; ++r) ex[r * 64 + ol] = {w[hw_br5(26 + r)].y, w[hw_br5(26 + r)].x};
            wave_lds_fence();
            const int olw = L == 1024 ? ol >> 1 : ol;
            const int mlw = (WD - olw) & (WD - 1);
            const int mol = L == 1024 ? 2 * mlw + (ol & 1) : mlw;
            const int first = olw == 0 ? 1 : 0;
#pragma unroll
            for (int q = 0; q < PW_NQ; ++q) {
                yp[q] = {w[hw_br5(q)].y, w[hw_br5(q)].x};
                const cx<double> m = ex[(5 - q + first) * 64 + mol];
                ym[q] = (q == 0 && first) ? yp[0] : m;
            }
            wave_lds_fence();
        }
        // The next item's samples are requested HERE, behind the second transform, and travel under the magnitudes and the
        // arg-max rounds.  Until the end of round 6 they were requested in front of the first transform: 44 registers more across
        // both transforms, which the 1024-point kernel does not have -- the compiler loaded four of them, WAITED for them
        // (s_waitcnt vmcnt(0) inside the prefetch, twice per lane class), spilled them and reloaded them at the top of the next
        // iteration: 36 bytes of scratch and four exposed memory latencies per iteration.
        advance();
        const bool more = __builtin_amdgcn_readfirstlane((int)live_here()) != 0;   // (lane 0: the even class's item; wave-uniform)
        it = item_here();
        fetch(it, xa, xb);
        hw_phase();
        // 4 |X_a[k]|^2 and 4 |X_b[k]|^2 (-inf from `half` on).  X[k] = conj(chirp[k]) y[k] and X[-k] = conj(chirp[k]) y[-k] (the
        // chirp is even); X_a = (X[k] + conj X[-k]) / 2, X_b = (X[k] - conj X[-k]) / 2i.  Times conj(chirp[k]) once more -- a
        // factor of modulus 1 -- the two are t +- conj(y[-k]) with t = conj(chirp[k])^2 y[k]: one complex product per bin; the
        // halves wait for the winner (a power of two commutes with the square root).
        double ma[PW_NQ], mb[PW_NQ];
        {
            const int ol = hw_opaque(lw);
#pragma unroll
            for (int q = 0; q < PW_NQ; ++q) {
                const cx<double> oc2 = oc_lds[q * WD + ol];
                const cx<double> t = cmul(yp[q], oc2);
                const cx<double> sa = {t.x + ym[q].x, t.y - ym[q].y}, sb = {t.x - ym[q].x, t.y + ym[q].y};
                const bool in = ol + WD * q < half;
                const double a2 = live_a ? fma(sa.x, sa.x, sa.y * sa.y) : 0.0, b2 = live_b ? fma(sb.x, sb.x, sb.y * sb.y) : 0.0;
                ma[q] = in ? a2 : -INFINITY;
                mb[q] = in ? b2 : -INFINITY;
            }
        }
        const bool has_item = cur_slot >= 0;
        for (int run = 0; run < runs; ++run) {
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                double* m = f ? mb : ma;
                // numpy argmax: first index of the maximum, NaN never wins
                double bv = -INFINITY;
#pragma unroll
                for (int q = 0; q < PW_NQ; ++q)
                    if (m[q] > bv) bv = m[q];
                const double wmax = pw_class_max<L>(bv, parity);
                const int olw = hw_opaque(lw);
                int key = 0x7fffffff;   // this lane's first bin that holds the maximum
#pragma unroll
                for (int q = PW_NQ - 1; q >= 0; --q) key = m[q] == wmax ? olw + WD * q : key;
                int idx = pw_class_min<L>(key, parity);
                if (!(wmax > -INFINITY)) idx = 0x7fffffff;
                int pc = -1;
                double val = 0.0;
                if (half > 0) {
                    const bool none = idx == 0x7fffffff;   // nothing compared greater than -inf: every magnitude is NaN
                    if (none) idx = 0;
                    // the bin's pitch class from the host's table (12 (log2(k val) - log2 440) + 69, rounded, modulo 12); -2: hz_to_note
                    // raises on NaN (ValueError) and on +-inf (OverflowError, e.g. the DC bin): the reference `continue`s: nothing is
                    // added and nothing is eliminated (prime_multif0.py:73-74)
                    pc = pc_lds[idx];
                    if (pc >= 0) {
                        val = none ? __builtin_nan("") : sqrt(wmax) * scale;   // mlab: np.abs(result) / window.sum()
                        if (run + 1 < runs) {   // (what the last round eliminates nobody looks at)
                            // f == k * max_f, exact comparison (:80).  Two bins' frequencies differ by `val`, far more than a
                            // rounding error: only bin k idx can compare equal (and does not always).  Its owner notes it.
                            const double max_f = (double)idx * c.val;
                            int zero = 0;   // bit q: this lane's bin lw + WD q goes
#pragma nounroll
                            for (int k = 1; k < elim; ++k) {
                                const int t = k * idx;
                                const bool eq = (double)t * c.val == (double)k * max_f;
                                if (eq && t < half && (t & (WD - 1)) == olw) zero |= 1 << (t / WD);
                            }
#pragma unroll
                            for (int q = 0; q < PW_NQ; ++q) m[q] = ((zero >> q) & 1) ? 0.0 : m[q];
                        }
                        // unicode-sharp quirk A.18 (MPX_NOTES_UNICODE): sharps land in a stray key and are lost, but the
                        // elimination above has happened; ASCII note names (librosa < 0.8) keep every pitch class
                        if (note_names == MPX_NOTES_UNICODE && ((0x54A >> pc) & 1)) pc = -1;
                    } else {
                        pc = -1;
                    }
                }
                if (lw == 0 && has_item && (f == 0 || cur_b)) {
                    out_pc[(cur_slot + f) * PRIME_MAX_RUNS + run] = pc;
                    out_val[(cur_slot + f) * PRIME_MAX_RUNS + run] = val;
                }
            }
        }
        if (!more) break;
    }
}

// one wave per clip: chroma[clip] = sum over its item slots.  Lane l adds up the slots l, l + 64, ... (twelve sums in
// registers: a slot holds up to PRIME_MAX_RUNS (pitch class, value) pairs), then lane n < 12 adds the 64 partial sums of
// pitch class n in lane order: a fixed order that depends on nothing but the clip's own slot count.  (Until round 5 twelve
// lanes walked all the slots one after the other: 0.85 ms per 4096 clips, 3.5 % of the method once its transforms took 24 ms.)
__global__ __launch_bounds__(64) void prime_sum_kernel(const long long* __restrict__ seg, int runs,
                                                       const int* __restrict__ pc, const double* __restrict__ val,
                                                       double* out) {
    __shared__ double part[64][13];
    const int lane = threadIdx.x;
    const long long s0 = seg[blockIdx.x], s1 = seg[blockIdx.x + 1];
    double acc[12];
#pragma unroll
    for (int n = 0; n < 12; ++n) acc[n] = 0.0;
    for (long long s = s0 + lane; s < s1; s += 64)
        for (int r = 0; r < runs; ++r) {
            const int p = pc[s * PRIME_MAX_RUNS + r];
            const double v = val[s * PRIME_MAX_RUNS + r];
#pragma unroll
            for (int n = 0; n < 12; ++n) acc[n] += p == n ? v : 0.0;
        }
#pragma unroll
    for (int n = 0; n < 12; ++n) part[lane][n] = acc[n];
    __syncthreads();
    if (lane < 12) {
        double t = 0.0;
        for (int l = 0; l < 64; ++l) t += part[l][lane];
        out[(long long)blockIdx.x * 12 + lane] = t;
    }
}

static void prime_host_fft(std::vector<cx<double>>& a) {
    const size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t len = 2; len <= n; len <<= 1)
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const long double ang = -2.0L * M_PIl * (long double)k / (long double)len;
                const cx<double> w = {(double)cosl(ang), (double)sinl(ang)};
                const cx<double> u = a[i + k], t = a[i + k + len / 2];
                const cx<double> v = {t.x * w.x - t.y * w.y, t.x * w.y + t.y * w.x};
                a[i + k] = {u.x + v.x, u.y + v.y};
                a[i + k + len / 2] = {u.x - v.x, u.y - v.y};
            }
}

// frequency held by register e of thread t after dif_fft_keep_last<L>
template <int L>
static int prime_reg_freq_t(int t, int e) {
    constexpr int RL = DifPlan<L>::radix(DifPlan<L>::n - 1);
    return dif_freq<L>(dif_last_pos<L>(t, e / RL, e % RL));
}
static int prime_reg_freq(int L, int t, int e) {
    return L == 1024 ? prime_reg_freq_t<1024>(t, e) : (L == 2048 ? prime_reg_freq_t<2048>(t, e) : prime_reg_freq_t<4096>(t, e));
}

struct PrimePlan {
    std::vector<PrimeCand> cands;  // host copy (pointers are device pointers)
    PrimeCand* d_cands = nullptr;
};

// Plans live in the context (host copy of the candidate records in ctx->host_blobs, device tables in
// ctx->owned) and die with it.
static int prime_plan(mpx_ctx* ctx, int fs, const mpx_prime_params& p, PrimePlan& plan) {
    const std::string key = "prime6_" + std::to_string(fs) + "_" + std::to_string(p.num_harmonic) + "_" +
                            std::to_string(p.num_octave);
    auto bit = ctx->host_blobs.find(key);
    if (bit != ctx->host_blobs.end()) {
        const size_t cnt = bit->second.size() / sizeof(PrimeCand);
        plan.cands.resize(cnt);
        memcpy(plan.cands.data(), bit->second.data(), cnt * sizeof(PrimeCand));
        plan.d_cands = (PrimeCand*)ctx->misc_plans[key][0];
        return MPX_OK;
    }
    const double c3 = 440.0 * std::pow(2.0, (48.0 - 69.0) / 12.0);
    std::map<int, void*> twl;
    for (int n = 0; n < 12; ++n) {
        const double note = c3 * std::pow(2.0, n / 12.0);
        for (int oct = 1; oct <= p.num_octave; ++oct)
            for (int h = 1; h <= p.num_harmonic; ++h) {
                const double f = note * oct * h;
                const int N = (int)((8 / f) * fs);  // prime_multif0.py:53
                // Only the lower half of the one-sided spectrum is looked at (prime_multif0.py:59-61): K = half outputs.  The
                // persistent kernel transforms two real frames at once and separates them by conjugate symmetry, which takes
                // the outputs -(K-1) .. K-1: the chirp-z convolution then spans chirp[-(N+K-2) .. K-1], a circular length of
                // N + 2K - 2 ~ 1.5 N (a full transform: 2N - 1).  Frames whose 1.5 N does not fit 4096 points but whose 1.25 N
                // does (2731 .. 3277 samples: the lowest candidates of 48 kHz input) run on the same kernel one frame at a
                // time; longer ones (up to 6553 samples, input above 53 kHz) on the workgroup-per-frame kernel.
                const int half = N >= 2 ? (N / 2 + 1) / 2 : 0;
                // frames above 6553 samples (input above ~107 kHz): the input decimated by R, R passes of the 8192-point
                // chirp-z (prime_kernel); a thread accumulates at most 8 bins: half <= 4096, i.e. frames up to 16 384 samples
                int R = 1;
                while (N >= 2 && (N + R - 1) / R + half - 1 > 8192 && R < 64) ++R;
                if (N < 2 || (N + R - 1) / R + half - 1 > 8192 || (R > 1 && half > 4096))
                    return set_error(ctx, MPX_EUNSUPPORTED,
                                     "prime-multiF0: frame of %d samples for candidate %.2f Hz (supported: 2..16384)", N, f);
                PrimeCand c;
                c.N = N;
                c.R = R;
                c.n1 = (N + R - 1) / R;
                c.coef = nullptr;
                if (R > 1) {
                    c.paired = 0;
                    c.L = 8192;
                    c.half = half;
                    c.val = 1.0 / (N * (1.0 / fs));
                    std::vector<double> win(N);
                    double wsum = 0.0;
                    for (int i = 0; i < N; ++i) win[i] = 0.5 - 0.5 * std::cos(2.0 * M_PI * i / (double)(N - 1));
                    for (int i = 0; i < N; ++i) wsum += win[i];
                    c.wsum = wsum;
                    const int J = c.n1 > half ? c.n1 : half;
                    std::vector<cx<double>> chirp(J), filt(c.L, cx<double>{0.0, 0.0}), coef((size_t)R * half);
                    for (long long i = 0; i < J; ++i) {   // exp(i pi R j^2 / N), the phase reduced exactly
                        const long long q = (long long)(((unsigned long long)R * (unsigned long long)(i * i)) % (unsigned long long)(2LL * N));
                        const long double ang = M_PIl * (long double)q / (long double)N;
                        chirp[i] = {(double)cosl(ang), (double)sinl(ang)};
                    }
                    for (int m = 0; m < half; ++m) filt[m] = chirp[m];              // k - m = 0 .. half-1
                    for (int m = 1; m < c.n1; ++m) filt[c.L - m] = chirp[m];        // k - m = -1 .. -(n1-1)
                    prime_host_fft(filt);
                    for (auto& v : filt) {
                        v.x /= c.L;
                        v.y /= c.L;
                    }
                    for (int r = 0; r < R; ++r)
                        for (int k = 0; k < half; ++k) {
                            const long double ang = -2.0L * M_PIl * (long double)(((long long)r * k) % N) / (long double)N;
                            const cx<double> w = {(double)cosl(ang), (double)sinl(ang)}, ch = chirp[k];
                            coef[(size_t)r * half + k] = {w.x * ch.x + w.y * ch.y, w.y * ch.x - w.x * ch.y};   // w * conj(chirp)
                        }
                    if (!twl.count(c.L)) {
                        std::vector<cx<double>> tw(c.L);
                        for (int j = 0; j < c.L; ++j) {
                            const long double ang = -2.0L * M_PIl * j / (long double)c.L;
                            tw[j] = {(double)cosl(ang), (double)sinl(ang)};
                        }
                        twl[c.L] = upload(ctx, tw.data(), tw.size() * sizeof(cx<double>));
                        if (!twl[c.L]) return MPX_ENOMEM;
                    }
                    c.tw = (const cx<double>*)twl[c.L];
                    c.win = (const double*)upload(ctx, win.data(), win.size() * sizeof(double));
                    c.chirp = (const cx<double>*)upload(ctx, chirp.data(), chirp.size() * sizeof(cx<double>));
                    c.bhat = (const cx<double>*)upload(ctx, filt.data(), filt.size() * sizeof(cx<double>));
                    c.coef = (const cx<double>*)upload(ctx, coef.data(), coef.size() * sizeof(cx<double>));
                    c.bhat_r = nullptr;
                    c.wv_wc = c.wv_fr = c.wv_oc = c.wv_theta = c.wv_tw2 = nullptr;
                c.wv_pc = nullptr;
                    if (!c.win || !c.chirp || !c.bhat || !c.coef) return MPX_ENOMEM;
                    plan.cands.push_back(c);
                    continue;
                }
                const int need2 = N + 2 * half - 2, need1 = N + half - 1;
                const bool paired = need2 <= 4096;
                const int need = paired ? need2 : need1;
                c.paired = paired ? 1 : 0;
                c.L = need > 4096 ? 8192 : (need <= 1024 ? 1024 : (need <= 2048 ? 2048 : 4096));
                c.half = half;
                c.val = 1.0 / (N * (1.0 / fs));
                std::vector<double> win(N);
                double wsum = 0.0;
                for (int i = 0; i < N; ++i) {  // numpy.hanning(N)
                    win[i] = N == 1 ? 1.0 : 0.5 - 0.5 * std::cos(2.0 * M_PI * i / (double)(N - 1));
                }
                // numpy's pairwise summation differs from a left-to-right sum by a few ulp; the value only
                // scales the magnitudes (1e-16 relative), never a comparison
                for (int i = 0; i < N; ++i) wsum += win[i];
                c.wsum = wsum;
                std::vector<cx<double>> chirp(N), filt(c.L, cx<double>{0.0, 0.0});
                for (long long i = 0; i < N; ++i) {
                    const long long q = (i * i) % (2LL * N);
                    const long double ang = M_PIl * (long double)q / (long double)N;
                    chirp[i] = {(double)cosl(ang), (double)sinl(ang)};
                }
                auto chirp_at = [&](long long i) {   // e^{i pi i^2 / N} for any index (phase reduced exactly)
                    const long long q = (i * i) % (2LL * N);
                    const long double ang = M_PIl * (long double)q / (long double)N;
                    return cx<double>{(double)cosl(ang), (double)sinl(ang)};
                };
                for (int m = 0; m < (half > 1 ? half : 1); ++m) filt[m] = chirp[m];   // chirp[k - n], k - n = 0 .. K-1
                const int neg = paired ? N + half - 2 : N - 1;                        // k - n = -1 .. -neg (the chirp is even)
                for (int m = 1; m <= neg; ++m) filt[c.L - m] = chirp_at(m);
                prime_host_fft(filt);
                for (auto& v : filt) {
                    v.x /= c.L;
                    v.y /= c.L;
                }
                if (!twl.count(c.L)) {
                    std::vector<cx<double>> tw(c.L);
                    for (int j = 0; j < c.L; ++j) {
                        const long double ang = -2.0L * M_PIl * j / (long double)c.L;
                        tw[j] = {(double)cosl(ang), (double)sinl(ang)};
                    }
                    twl[c.L] = upload(ctx, tw.data(), tw.size() * sizeof(cx<double>));
                    if (!twl[c.L]) return MPX_ENOMEM;
                }
                c.tw = (const cx<double>*)twl[c.L];
                c.win = (const double*)upload(ctx, win.data(), win.size() * sizeof(double));
                c.chirp = (const cx<double>*)upload(ctx, chirp.data(), chirp.size() * sizeof(cx<double>));
                c.bhat = (const cx<double>*)upload(ctx, filt.data(), filt.size() * sizeof(cx<double>));
                c.bhat_r = nullptr;
                if (c.L <= 4096) {   // prime_pers_kernel multiplies the filter onto the registers the forward DIF leaves: [e][tid]
                    std::vector<cx<double>> fr(c.L);
                    for (int t = 0; t < c.L / 8; ++t)
                        for (int e = 0; e < 8; ++e) fr[(size_t)e * (c.L / 8) + t] = filt[prime_reg_freq(c.L, t, e)];
                    c.bhat_r = (const cx<double>*)upload(ctx, fr.data(), fr.size() * sizeof(cx<double>));
                    if (!c.bhat_r) return MPX_ENOMEM;
                }
                if (!c.win || !c.chirp || !c.bhat) return MPX_ENOMEM;
                c.wv_wc = c.wv_fr = c.wv_oc = c.wv_theta = c.wv_tw2 = nullptr;
                if (c.L <= 2048) {   // prime_wave_kernel's tables (tests/test_prime_wave_algorithm.py restates them)
                    const int WD = c.L / 32;
                    if (PW_NR * WD < N || PW_NQ * WD < half || !c.paired)
                        return set_error(ctx, MPX_EUNSUPPORTED, "prime-multiF0: frame of %d samples does not fit its wave class", N);
                    std::vector<cx<double>> wc((size_t)PW_NR * WD, cx<double>{0.0, 0.0}), oc((size_t)PW_NQ * WD, cx<double>{0.0, 0.0}),
                        fr((size_t)32 * WD);
                    for (int i = 0; i < N; ++i) wc[i] = {win[i] * chirp[i].x, -(win[i] * chirp[i].y)};     // [n1][lane]: n = lane + WD n1
                    std::vector<int> pcs((size_t)PW_NQ * WD, -2);
                    for (long long k = 0; k < half; ++k) {                                                   // [q][lane]: k = lane + WD q
                        const long long q2 = (2 * k * k) % (2LL * N);
                        const long double ang = -M_PIl * (long double)q2 / (long double)N;
                        oc[k] = {(double)cosl(ang), (double)sinl(ang)};                                      // conj(chirp[k])^2
                        const double max_f = (double)k * c.val;
                        const double midi = 12.0 * (std::log2(max_f) - std::log2(440.0)) + 69.0;
                        if (midi == midi && !std::isinf(midi)) {
                            const long long note = (long long)std::nearbyint(midi);
                            pcs[k] = (int)(((note % 12) + 12) % 12);
                        }
                    }
                    for (int p = 0; p < 32; ++p)
                        for (int l = 0; l < WD; ++l)
                            fr[(size_t)p * WD + l] = c.L == 1024 ? filt[l + 32 * hw_br5(p)] : filt[(l >> 1) + 32 * hw_br5(p) + 1024 * (l & 1)];
                    if (!twl.count(-1)) {
                        std::vector<cx<double>> th(160), tw2(32 * 33, cx<double>{1.0, 0.0});
                        for (int st = 0; st < 5; ++st)
                            for (int k1 = 0; k1 < 32; ++k1) {
                                const long double ang = -2.0L * M_PIl * (long double)(k1 * (16 >> st)) / 1024.0L;
                                th[32 * st + k1] = {(double)cosl(ang), (double)sinl(ang)};
                            }
                        for (int p = 0; p < 32; ++p)
                            for (int k1 = 0; k1 < 32; ++k1) {
                                const long double ang = -2.0L * M_PIl * (long double)(k1 + 32 * hw_br5(p)) / 2048.0L;
                                tw2[33 * p + k1] = {(double)cosl(ang), (double)sinl(ang)};
                            }
                        twl[-1] = upload(ctx, th.data(), th.size() * sizeof(cx<double>));
                        twl[-2] = upload(ctx, tw2.data(), tw2.size() * sizeof(cx<double>));
                        if (!twl[-1] || !twl[-2]) return MPX_ENOMEM;
                    }
                    c.wv_theta = (const cx<double>*)twl[-1];
                    c.wv_tw2 = (const cx<double>*)twl[-2];
                    c.wv_wc = (const cx<double>*)upload(ctx, wc.data(), wc.size() * sizeof(cx<double>));
                    c.wv_oc = (const cx<double>*)upload(ctx, oc.data(), oc.size() * sizeof(cx<double>));
                    c.wv_fr = (const cx<double>*)upload(ctx, fr.data(), fr.size() * sizeof(cx<double>));
                    c.wv_pc = (const int*)upload(ctx, pcs.data(), pcs.size() * sizeof(int));
                    if (!c.wv_wc || !c.wv_oc || !c.wv_fr || !c.wv_pc) return MPX_ENOMEM;
                }
                plan.cands.push_back(c);
            }
    }
    plan.d_cands = (PrimeCand*)upload(ctx, plan.cands.data(), plan.cands.size() * sizeof(PrimeCand));
    if (!plan.d_cands) return MPX_ENOMEM;
    std::vector<unsigned char> blob(plan.cands.size() * sizeof(PrimeCand));
    memcpy(blob.data(), plan.cands.data(), blob.size());
    ctx->host_blobs[key] = std::move(blob);
    ctx->misc_plans[key] = {plan.d_cands};
    return MPX_OK;
}

template <int L, int T>
static void prime_launch(const float* d_sig, const PrimeItem* d_items, size_t count, const PrimeCand* d_cands, int runs,
                         int elim, int note_names, int* d_pc, double* d_val, hipStream_t st, int per_clip, long long clip_len,
                         long long clip_slots) {
    if (!count) return;
    const size_t lds = sizeof(cx<double>) * lds_slots(L);  // the magnitudes alias the upper half of the buffer
    auto kern = prime_kernel<L, T>;
    if (lds > 48 * 1024) hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)count), dim3(T), lds, st, d_sig, d_items, d_cands, runs, elim, note_names, d_pc, d_val, per_clip, clip_len, clip_slots);
}

template <int L>
static int prime_pers_launch(mpx_ctx* ctx, const float* d_sig, const PrimeItem* d_items, const PrimeWork* d_work, size_t groups,
                             const PrimeCand* d_cands, int runs, int elim, int note_names, int* d_pc, double* d_val, hipStream_t st,
                             int uniform_clips, long long clip_len, long long clip_slots) {
    if (!groups) return MPX_OK;
    const size_t lds = sizeof(cx<double>) * L;
    auto kern = prime_pers_kernel<L>;
    if (lds > 48 * 1024) MPX_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)groups), dim3(L / 8), lds, st, d_sig, d_items, d_work, d_cands, runs, elim, note_names,
                       d_pc, d_val, uniform_clips, clip_len, clip_slots);
    return MPX_OK;
}
template <int L, int WAVES>
static int prime_wave_launch(mpx_ctx* ctx, const float* d_sig, const PrimeItem* d_items, const PrimeWork* d_work, size_t groups,
                             const PrimeCand* d_cands, int runs, int elim, int note_names, int* d_pc, double* d_val, hipStream_t st,
                             int uniform_clips, long long clip_len, long long clip_slots) {
    if (!groups) return MPX_OK;
    const size_t lds = (size_t)pw_table_elems<L>() * 16 + (size_t)WAVES * HW_XBUF;
    auto kern = prime_wave_kernel<L, WAVES>;
    MPX_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)groups), dim3(WAVES * 64), lds, st, d_sig, d_items, d_work, d_cands, runs, elim, note_names,
                       d_pc, d_val, uniform_clips, clip_len, clip_slots);
    return MPX_OK;
}
// resident workgroups of prime_pers_kernel<L> on the whole device (cached per context)
template <int L>
static int prime_pers_slots(mpx_ctx* ctx) {
    const std::string key = "prime_pers_" + std::to_string(L);
    auto it = ctx->occupancy.find(key);
    if (it != ctx->occupancy.end()) return it->second;
    const size_t lds = sizeof(cx<double>) * L;
    auto kern = prime_pers_kernel<L>;
    if (lds > 48 * 1024) hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, L / 8, lds) != hipSuccess || occ < 1) occ = 1;
    return ctx->occupancy[key] = occ * ctx->num_cus;
}

// signals: packed clips on the HOST; offsets[C+1]; out: [C,12] on the host
// dev_io: `signals` is DEVICE memory used in place and chroma_sums a device buffer; the kernels are only enqueued on
// `stream` (the host-built item lists are uploaded and waited for first).
int prime_run_host(mpx_ctx* ctx, const float* signals, const int64_t* offsets, int num_clips, int fs,
                   const mpx_prime_params* params, double* chroma_sums, bool dev_io, hipStream_t stream) {
    mpx_prime_params p = params ? *params : mpx_prime_params{1, 2, 5, 2, MPX_NOTES_UNICODE};
    if (p.num_harmonic < 1 || p.num_octave < 1 || p.num_harmonic * p.num_octave > 64 || p.harmonic_multiples_elim < 1 ||
        p.harmonic_multiples_elim > 64 || p.harmonic_elim_runs < 0 || p.harmonic_elim_runs > PRIME_MAX_RUNS)
        return set_error(ctx, MPX_EINVAL, "bad prime-multiF0 params");
    if (p.note_names != MPX_NOTES_UNICODE && p.note_names != MPX_NOTES_ASCII)
        return set_error(ctx, MPX_EINVAL, "prime-multiF0: unknown note_names %d", p.note_names);
    if (fs <= 0) return set_error(ctx, MPX_EINVAL, "fs must be positive");
    PrimePlan plan_storage;
    int rc = prime_plan(ctx, fs, p, plan_storage);
    if (rc) return rc;
    PrimePlan* plan = &plan_storage;
    const int64_t total = offsets[num_clips];
    // slots in the reference's loop order (clip, candidate, frame); items grouped by CANDIDATE (all clips of a candidate
    // together: a persistent workgroup is bound to one candidate), candidates grouped by chirp-z class
    std::vector<std::vector<PrimeItem>> by_cand(plan->cands.size());
    std::vector<long long> seg(1, 0);
    long long slot = 0;
    // A batch of equal-length clips (a corpus) needs the item list of ONE clip: the kernel derives the others.  Building
    // and uploading 1700 items x 24 B for each of 4096 clips was half of the call's time.
    bool uniform = num_clips > 1;
    for (int cidx = 1; cidx <= num_clips && uniform; ++cidx)
        uniform = offsets[cidx] - offsets[cidx - 1] == offsets[1] - offsets[0];
    const int built_clips = uniform ? 1 : num_clips;
    for (int cidx = 0; cidx < built_clips; ++cidx) {
        const int64_t len = offsets[cidx + 1] - offsets[cidx];
        if (len < 0) return set_error(ctx, MPX_EINVAL, "offsets must be non-decreasing");
        for (size_t k = 0; k < plan->cands.size(); ++k) {
            const PrimeCand& c = plan->cands[k];
            const int64_t nf = len <= 0 ? 0 : (len + c.N - 1) / c.N;
            const int step = c.paired ? 2 : 1;      // prime_pers_kernel: frames 2p and 2p + 1 of a clip travel together
            for (int64_t f = 0; f < nf; f += step) {
                const int64_t s = f * c.N, left = len - s;
                PrimeItem it;
                it.start = offsets[cidx] + s;
                it.valid = (int)(left >= c.N ? c.N : left);
                it.cand = (int)k;
                it.slot = slot;
                it.valid_b = 0;
                it.pad = 0;
                if (step == 2 && f + 1 < nf) {
                    const int64_t left_b = left - c.N;
                    it.valid_b = (int)(left_b >= c.N ? c.N : left_b);
                }
                slot += it.valid_b ? 2 : 1;
                by_cand[k].push_back(it);
            }
        }
        seg.push_back(slot);
    }
    auto class_of = [](int L) { return L == 1024 ? 0 : (L == 2048 ? 1 : (L == 4096 ? 2 : 3)); };
    std::vector<PrimeItem> items[4];
    std::vector<int> cand_item0(plan->cands.size(), 0);
    for (size_t k = 0; k < plan->cands.size(); ++k) {
        auto& v = items[class_of(plan->cands[k].L)];
        if (by_cand[k].size() > (size_t)0x7fff0000u)   // prime_wave_kernel walks a candidate's items with a 32-bit index + stride
            return set_error(ctx, MPX_EUNSUPPORTED, "prime-multiF0: %zu frame pairs of one candidate frequency in one call", by_cand[k].size());
        cand_item0[k] = (int)v.size();
        v.insert(v.end(), by_cand[k].begin(), by_cand[k].end());
    }
    const long long clip_slots = uniform ? slot : 0, clip_len = uniform ? offsets[1] - offsets[0] : 0;
    if (uniform) {
        if (clip_len < 0) return set_error(ctx, MPX_EINVAL, "offsets must be non-decreasing");
        for (int cidx = 1; cidx < num_clips; ++cidx) seg.push_back(clip_slots * (cidx + 1));
        slot = clip_slots * num_clips;
    }
    hipStream_t st = stream ? stream : ctx->stream;
    // (round 6: the batch entry point reads clips that are already in HBM in place too, like mpx_*_batch of the framed methods --
    //  until then it copied them into the context's buffer: 0.9 GB more to allocate and 0.3 ms per 4096 two-second clips)
    const bool in_dev = dev_io || (total && samples_on_device(ctx, signals));
    if (!in_dev && (rc = ensure(ctx, ctx->d_signal, (size_t)(total ? total : 1) * sizeof(float)))) return rc;
    const size_t nitems = (size_t)slot;
    size_t item_bytes = 0;
    for (auto& v : items) item_bytes += v.size() * sizeof(PrimeItem);
    ctx->batch_layout.clear();   // d_desc / d_offsets are about to hold this call's tables (method_batch's cache)
    if ((rc = ensure(ctx, ctx->d_desc, item_bytes + 64))) return rc;
    if ((rc = ensure(ctx, ctx->d_offsets, seg.size() * sizeof(long long)))) return rc;
    if ((rc = ensure(ctx, ctx->d_ws0, (nitems + 1) * PRIME_MAX_RUNS * (sizeof(int) + sizeof(double)) + 64))) return rc;
    if ((rc = ensure(ctx, ctx->d_sum, (size_t)(num_clips ? num_clips : 1) * 12 * sizeof(double)))) return rc;
    if (!in_dev && total && (rc = stage_h2d(ctx, ctx->d_signal.p, signals, (size_t)total * sizeof(float), st))) return rc;
    const float* d_in = in_dev ? signals : (const float*)ctx->d_signal.p;   // the kernel reads valid samples only
    MPX_HIP(ctx, hipMemcpyAsync(ctx->d_offsets.p, seg.data(), seg.size() * sizeof(long long), hipMemcpyHostToDevice, st));
    double* d_val = (double*)ctx->d_ws0.p;
    int* d_pc = (int*)(d_val + (nitems + 1) * PRIME_MAX_RUNS);
    char* d_items = (char*)ctx->d_desc.p;
    size_t off = 0;
    for (int cls = 0; cls < 4; ++cls) {
        const size_t bytes = items[cls].size() * sizeof(PrimeItem);
        if (bytes) MPX_HIP(ctx, hipMemcpyAsync(d_items + off, items[cls].data(), bytes, hipMemcpyHostToDevice, st));
        off += bytes;
    }
    // work table of the persistent kernel: the resident workgroups of a class are shared out among its candidates in
    // proportion to their items (equal cost inside a class), so that every workgroup walks about the same number of frames
    // (prime_wave_kernel, classes 0 and 1: one workgroup per CU -- its tables and its waves' buffers fill the LDS -- and a
    //  worker is a WAVE: workgroup g of a candidate holds the workers g WAVES ... g WAVES + WAVES - 1, each taking one item
    //  (two in the 1024-point class) per iteration)
    std::vector<PrimeWork> work[3];
#ifdef MPX_DEV_KNOBS
    const bool wave_path = !dev_env_on("MPX_PRIME_PERS");
#else
    constexpr bool wave_path = true;
#endif
    int wave_waves[2] = {PW_WAVES_1024, PW_WAVES_2048};
    const int wave_per[2] = {2, 1};
#ifdef MPX_DEV_KNOBS   // development: other workgroup sizes (scripts/dev/prime_time.py)
    {
        const int w0 = dev_env_int("MPX_PRIME_WAVES_1024", PW_WAVES_1024), w1 = dev_env_int("MPX_PRIME_WAVES_2048", PW_WAVES_2048);
        if (w0 == 4 || w0 == 6) wave_waves[0] = w0;
        if (w1 == 2 || w1 == 3) wave_waves[1] = w1;
    }
#endif
    int slots_of[3] = {ctx->num_cus, ctx->num_cus, prime_pers_slots<4096>(ctx)};
#ifdef MPX_DEV_KNOBS
    if (!wave_path) {
        slots_of[0] = prime_pers_slots<1024>(ctx);
        slots_of[1] = prime_pers_slots<2048>(ctx);
    }
#endif
    for (int cls = 0; cls < 3; ++cls) {
        const long long mult = uniform ? num_clips : 1;
        const long long class_items = (long long)items[cls].size() * mult;
        if (!class_items) continue;
        const bool wv = wave_path && cls < 2;
        for (size_t k = 0; k < plan->cands.size(); ++k) {
            if (class_of(plan->cands[k].L) != cls || by_cand[k].empty()) continue;
            const long long mine = (long long)by_cand[k].size() * mult;
            long long w = (long long)((double)slots_of[cls] * (double)mine / (double)class_items + 0.5);
            const long long per_group = wv ? wave_waves[cls] * wave_per[cls] : 1;   // items a workgroup takes per iteration
            const long long cap = (mine + per_group - 1) / per_group;
            w = w < 1 ? 1 : (w > cap ? cap : w);
            for (long long i = 0; i < w; ++i)
                work[cls].push_back(wv ? PrimeWork{(int)k, (int)(i * wave_waves[cls]), (int)(w * wave_waves[cls]), cand_item0[k], (int)by_cand[k].size()}
                                       : PrimeWork{(int)k, (int)i, (int)w, cand_item0[k], (int)by_cand[k].size()});
        }
    }
    const size_t work_bytes = (work[0].size() + work[1].size() + work[2].size()) * sizeof(PrimeWork);
    if ((rc = ensure(ctx, ctx->d_ws1, work_bytes + 64))) return rc;
    {
        size_t woff = 0;
        for (int cls = 0; cls < 3; ++cls) {
            const size_t bytes = work[cls].size() * sizeof(PrimeWork);
            if (bytes) MPX_HIP(ctx, hipMemcpyAsync((char*)ctx->d_ws1.p + woff, work[cls].data(), bytes, hipMemcpyHostToDevice, st));
            woff += bytes;
        }
    }
    if (dev_io) MPX_HIP(ctx, hipStreamSynchronize(st));   // seg, items and work are host vectors of this call
    off = 0;
    size_t woff = 0;
    prof_mark(ctx, st, "prime_kernel");
    const bool class_marks = dev_env_on("MPX_PRIME_CLASS_MARKS");   // development: a profile entry per chirp-z class
    static const char* const class_names[4] = {"prime_kernel_1024", "prime_kernel_2048", "prime_kernel_4096", "prime_kernel_8192"};
    for (int cls = 0; cls < 4; ++cls) {
        if (class_marks) prof_mark(ctx, st, class_names[cls]);
        const size_t bytes = items[cls].size() * sizeof(PrimeItem);
        const PrimeItem* di = (const PrimeItem*)(d_items + off);
        const int uclips = uniform ? num_clips : 0;
        if (cls < 3) {
            const PrimeWork* dw = (const PrimeWork*)((char*)ctx->d_ws1.p + woff);
            const size_t groups = work[cls].size();
#define PW_ARGS ctx, d_in, di, dw, groups, plan->d_cands, p.harmonic_elim_runs, p.harmonic_multiples_elim, p.note_names, d_pc, d_val, st, uclips, clip_len, clip_slots
            if (cls == 0 && wave_path && wave_waves[0] == PW_WAVES_1024) rc = prime_wave_launch<1024, PW_WAVES_1024>(PW_ARGS);
            if (cls == 1 && wave_path && wave_waves[1] == PW_WAVES_2048) rc = prime_wave_launch<2048, PW_WAVES_2048>(PW_ARGS);
#ifdef MPX_DEV_KNOBS
            if (cls == 0 && wave_path && wave_waves[0] == 4) rc = prime_wave_launch<1024, 4>(PW_ARGS);
            if (cls == 0 && wave_path && wave_waves[0] == 6) rc = prime_wave_launch<1024, 6>(PW_ARGS);
            if (cls == 1 && wave_path && wave_waves[1] == 2) rc = prime_wave_launch<2048, 2>(PW_ARGS);
            if (cls == 1 && wave_path && wave_waves[1] == 3) rc = prime_wave_launch<2048, 3>(PW_ARGS);
#endif
#undef PW_ARGS
#ifdef MPX_DEV_KNOBS   // round 3's kernel for these classes: A / B only (MPX_PRIME_PERS=1)
            if (cls == 0 && !wave_path) rc = prime_pers_launch<1024>(ctx, d_in, di, dw, groups, plan->d_cands, p.harmonic_elim_runs, p.harmonic_multiples_elim, p.note_names, d_pc, d_val, st, uclips, clip_len, clip_slots);
            if (cls == 1 && !wave_path) rc = prime_pers_launch<2048>(ctx, d_in, di, dw, groups, plan->d_cands, p.harmonic_elim_runs, p.harmonic_multiples_elim, p.note_names, d_pc, d_val, st, uclips, clip_len, clip_slots);
#endif
            if (cls == 2) rc = prime_pers_launch<4096>(ctx, d_in, di, dw, groups, plan->d_cands, p.harmonic_elim_runs, p.harmonic_multiples_elim, p.note_names, d_pc, d_val, st, uclips, clip_len, clip_slots);
            if (rc) return rc;
            woff += groups * sizeof(PrimeWork);
        } else {   // chirp-z on 8192 points (frames above 3277 samples: input rates above 53 kHz): a workgroup per item, Stockham engine
            const int per_clip = uniform ? (int)items[cls].size() : 0;
            const size_t count = uniform ? items[cls].size() * (size_t)num_clips : items[cls].size();
            prime_launch<8192, 512>(d_in, di, count, plan->d_cands, p.harmonic_elim_runs, p.harmonic_multiples_elim, p.note_names, d_pc, d_val, st, per_clip, clip_len, clip_slots);
        }
        off += bytes;
    }
    prof_mark(ctx, st, "prime_sum_kernel");
    if (num_clips)
        hipLaunchKernelGGL(prime_sum_kernel, dim3(num_clips), dim3(64), 0, st, (const long long*)ctx->d_offsets.p,
                           p.harmonic_elim_runs, d_pc, d_val, dev_io ? chroma_sums : (double*)ctx->d_sum.p);
    prof_mark(ctx, st, nullptr);
    MPX_HIP(ctx, hipGetLastError());
    if (dev_io) return MPX_OK;
    if (num_clips)
        MPX_HIP(ctx, hipMemcpyAsync(chroma_sums, ctx->d_sum.p, (size_t)num_clips * 12 * sizeof(double), hipMemcpyDeviceToHost, st));
    // the item vectors are read by the async copies above: wait before they go out of scope
    MPX_HIP(ctx, hipStreamSynchronize(st));
    return MPX_OK;
}

}  // namespace mpx
