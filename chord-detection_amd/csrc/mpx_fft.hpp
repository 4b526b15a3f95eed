// LDS-resident power-of-two complex FFT for one workgroup (gfx950 / CDNA4).
//
// One workgroup of T threads transforms M complex points that live in LDS.
// Every thread owns EPT = M/T points in registers during a pass; a pass is a
// Stockham autosort step of radix R in {2,4,8,16} (natural order in, natural
// order out, no bit reversal), so the data makes log_R(M) round trips through
// LDS and none through HBM.  Passes are fully unrolled at compile time; twiddles
// come from one W_M table (fp64-exact, built on the host) that stays in L2.
//
// No reference counterpart: the reference calls numpy.fft (pocketfft) at
// esacf.py:103-105 and harmonic_energy.py:43.  No rocFFT / hipFFT is used.
#pragma once
#include <hip/hip_runtime.h>

namespace mpx {

template <typename T>
struct cx {
    T x, y;
};

template <typename T>
__device__ __forceinline__ cx<T> cadd(cx<T> a, cx<T> b) { return {a.x + b.x, a.y + b.y}; }
template <typename T>
__device__ __forceinline__ cx<T> csub(cx<T> a, cx<T> b) { return {a.x - b.x, a.y - b.y}; }
template <typename T>
__device__ __forceinline__ cx<T> cmul(cx<T> a, cx<T> b) {
    return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
// multiply by -i  (forward-transform quarter turn)
template <typename T>
__device__ __forceinline__ cx<T> mul_mi(cx<T> a) { return {a.y, -a.x}; }

// ---------------------------------------------------------------- small DFTs
// In-register forward DFT of R points, natural order in and out.
template <int R, typename T>
struct SmallDft;

template <typename T>
struct SmallDft<2, T> {
    static __device__ __forceinline__ void run(cx<T>* v) {
        cx<T> a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    }
};

template <typename T>
struct SmallDft<4, T> {
    static __device__ __forceinline__ void run(cx<T>* v) {
        cx<T> t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
        cx<T> t2 = cadd(v[1], v[3]), t3 = mul_mi(csub(v[1], v[3]));
        v[0] = cadd(t0, t2);
        v[2] = csub(t0, t2);
        v[1] = cadd(t1, t3);
        v[3] = csub(t1, t3);
    }
};

template <typename T>
struct SmallDft<8, T> {
    static __device__ __forceinline__ void run(cx<T>* v) {
        const T h = (T)0.70710678118654752440;
        cx<T> e[4] = {v[0], v[2], v[4], v[6]};
        cx<T> o[4] = {v[1], v[3], v[5], v[7]};
        SmallDft<4, T>::run(e);
        SmallDft<4, T>::run(o);
        // o[k] *= W8^k
        o[1] = {h * (o[1].x + o[1].y), h * (o[1].y - o[1].x)};
        o[2] = mul_mi(o[2]);
        o[3] = {h * (o[3].y - o[3].x), -h * (o[3].x + o[3].y)};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = cadd(e[k], o[k]);
            v[k + 4] = csub(e[k], o[k]);
        }
    }
};

template <typename T>
struct SmallDft<16, T> {
    static __device__ __forceinline__ void run(cx<T>* v) {
        // W16^k, k = 1..7 (cos, -sin)
        const T c1 = (T)0.92387953251128675613, s1 = (T)0.38268343236508977173;
        const T h = (T)0.70710678118654752440;
        cx<T> e[8], o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            e[k] = v[2 * k];
            o[k] = v[2 * k + 1];
        }
        SmallDft<8, T>::run(e);
        SmallDft<8, T>::run(o);
        o[1] = cmul(o[1], cx<T>{c1, -s1});
        o[2] = {h * (o[2].x + o[2].y), h * (o[2].y - o[2].x)};
        o[3] = cmul(o[3], cx<T>{s1, -c1});
        o[4] = mul_mi(o[4]);
        o[5] = cmul(o[5], cx<T>{-s1, -c1});
        o[6] = {h * (o[6].y - o[6].x), -h * (o[6].x + o[6].y)};
        o[7] = cmul(o[7], cx<T>{-c1, -s1});
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[k] = cadd(e[k], o[k]);
            v[k + 8] = csub(e[k], o[k]);
        }
    }
};

// ---------------------------------------------------------------- LDS layout
// Logical point i lives at physical slot i + (i >> 4): one pad slot after every
// 16 points.  A Stockham pass writes with a lane stride of R points (R*16 B in
// fp64: every lane of a ds_write_b128 group on the same banks); the pad turns
// that stride into R+1 (R=16) / R+0.5 slots and spreads the group over distinct
// 16-byte bank slots.  Reads (lane stride 1) stay contiguous.
__host__ __device__ __forceinline__ constexpr int lds_slot(int i) { return i + (i >> 4); }
constexpr int lds_slots(int m) { return m + (m >> 4); }

// ---------------------------------------------------------------- radix plans
constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }

// Radix sequence for M points when a thread holds EPT of them: as many passes of
// radix min(EPT,16) as fit, then one smaller pass for the remainder.
template <int M, int EPT>
struct Plan {
    static constexpr int MAXR = EPT < 16 ? EPT : 16;
    static constexpr int LR = ilog2(MAXR);
    static constexpr int LM = ilog2(M);
    static constexpr int full = LM / LR, rem = LM % LR;
    static constexpr int n = full + (rem ? 1 : 0);
    __host__ __device__ __forceinline__ static constexpr int radix(int i) {
        return i < full ? MAXR : ((i == full && rem) ? (1 << rem) : 1);
    }
    // product of the radices of passes 0..i-1 (no recursion: see DifPlan)
    __host__ __device__ __forceinline__ static constexpr int done(int i) {
        return i <= full ? (1 << (LR * i)) : (1 << (LR * full + rem));
    }
};

// One Stockham pass of radix R over the M points in `buf` (LDS).
//   P    = product of the radices of the passes already done
//   tw   = W_M table: tw[j] = exp(-2*pi*i*j/M), j in [0, M)
//   regs = EPT-element per-thread register file (in: ignored unless FROM_REGS;
//          out: left holding the pass result when TO_REGS)
// FROM_REGS: the pass input is already in `regs` in read order
//            (regs[b*R + r] = point (tid + b*T) + r*(M/R)); skips the LDS read.
// TO_REGS:   skip the LDS write; regs[b*R + r] = output index
//            ((i/P)*P*R + i%P) + r*P for butterfly i = tid + b*T.
template <int M, int T, int R, int P, bool FROM_REGS, bool TO_REGS, typename Real>
__device__ __forceinline__ void stockham_pass(cx<Real>* buf, const cx<Real>* __restrict__ tw,
                                              cx<Real>* regs, int tid) {
    constexpr int EPT = M / T;
    constexpr int NB = M / R;    // butterflies in this pass
    constexpr int BPT = EPT / R; // butterflies per thread
    static_assert(BPT >= 1, "radix larger than the per-thread register file");
    static_assert(NB % T == 0, "butterflies must divide evenly over threads");
    static_assert(NB % 16 == 0, "read stride must keep the pad phase (slot(i + r*NB) = slot(i) + r*slot(NB))");
    if (!FROM_REGS) {
#pragma unroll
        for (int b = 0; b < BPT; ++b) {
            const int i = tid + b * T;
#pragma unroll
            for (int r = 0; r < R; ++r) regs[b * R + r] = buf[lds_slot(i) + r * lds_slots(NB)];
        }
        __syncthreads();  // everyone has read before anyone overwrites
    }
#pragma unroll
    for (int b = 0; b < BPT; ++b) {
        const int i = tid + b * T;
        cx<Real>* v = regs + b * R;
        if (P > 1) {
            // one table load (W_{PR}^k) per butterfly; the other R-2 twiddles are its powers,
            // built with multiplication depth <= 4 (error ~1e-15 in fp64)
            const int k = i & (P - 1);
            constexpr int STRIDE = M / (P * R);
            cx<Real> w[R];
            w[1] = tw[k * STRIDE];
            // Callers loop over frames with this pass inlined: the table load is loop invariant and
            // gets hoisted (good: 1 register pair per butterfly), but the R-2 powers must NOT be, or
            // they pin ~4*(R-2) VGPRs per pass and halve occupancy.  Make the base opaque here.
            asm volatile("" : "+v"(w[1].x), "+v"(w[1].y));
#pragma unroll
            for (int r = 2; r < R; ++r) {
                const int hb = 1 << (31 - __builtin_clz(r));  // highest power of two <= r
                w[r] = (r == hb) ? cmul(w[r / 2], w[r / 2]) : cmul(w[hb], w[r - hb]);
            }
#pragma unroll
            for (int r = 1; r < R; ++r) v[r] = cmul(v[r], w[r]);
        }
        SmallDft<R, Real>::run(v);
        if (!TO_REGS) {
            const int k = i & (P - 1);
            const int j = (i - k) * R + k;
            // slot(j + r*P) = slot(j) + slot(r*P): j's low 4 bits never carry into the pad phase
            // (P = 1: j = R*i, r < R <= 16;  P = 8: j & 15 = i & 7 < 8;  P >= 16: r*P is a multiple of 16)
            static_assert(P == 1 || P == 8 || P % 16 == 0, "unsupported pass order for padded addressing");
            static_assert(P != 1 || R == 16 || R == 8, "first pass must be radix 8 or 16");
            cx<Real>* dst = buf + lds_slot(j);
#pragma unroll
            for (int r = 0; r < R; ++r) dst[lds_slot(r * P)] = v[r];
        }
    }
    if (!TO_REGS) __syncthreads();
}

// Forward FFT of the M points in `buf` (LDS).  If FIRST_FROM_REGS the first pass
// takes its input from `regs` (see stockham_pass) and `buf` need not be
// initialised; the caller must have issued a __syncthreads() since the last
// read of `buf`.  Result is in `buf`, natural order, all threads synchronised.
template <int M, int T, int I, bool FIRST_FROM_REGS, typename Real>
__device__ __forceinline__ void fft_passes(cx<Real>* buf, const cx<Real>* __restrict__ tw, cx<Real>* regs,
                                           int tid) {
    using PL = Plan<M, M / T>;
    if constexpr (I < PL::n) {
        stockham_pass<M, T, PL::radix(I), PL::done(I), (I == 0 && FIRST_FROM_REGS), false, Real>(buf, tw, regs, tid);
        fft_passes<M, T, I + 1, FIRST_FROM_REGS, Real>(buf, tw, regs, tid);
    }
}

// Same, but the last pass leaves its outputs in `regs` instead of LDS:
// regs[e] holds Z[last_pass_index<M,T>(tid, e)].  `buf` is free after the call
// returns only once the caller synchronises.
template <int M, int T, int I, bool FIRST_FROM_REGS, typename Real>
__device__ __forceinline__ void fft_passes_keep(cx<Real>* buf, const cx<Real>* __restrict__ tw, cx<Real>* regs,
                                                int tid) {
    using PL = Plan<M, M / T>;
    if constexpr (I < PL::n) {
        stockham_pass<M, T, PL::radix(I), PL::done(I), (I == 0 && FIRST_FROM_REGS), (I == PL::n - 1), Real>(
            buf, tw, regs, tid);
        fft_passes_keep<M, T, I + 1, FIRST_FROM_REGS, Real>(buf, tw, regs, tid);
    }
}

template <int M, int T, bool FIRST_FROM_REGS, typename Real>
__device__ __forceinline__ void fft_lds_keep_last(cx<Real>* buf, const cx<Real>* __restrict__ tw,
                                                  cx<Real>* regs, int tid) {
    fft_passes_keep<M, T, 0, FIRST_FROM_REGS, Real>(buf, tw, regs, tid);
}

// Output index held by regs[e] after fft_lds_keep_last (last pass: P*R = M).
template <int M, int T>
__device__ __forceinline__ int last_pass_index(int tid, int e) {
    using PL = Plan<M, M / T>;
    constexpr int R = PL::radix(PL::n - 1);
    constexpr int P = PL::done(PL::n - 1);
    const int b = e / R, r = e % R;
    return tid + b * T + r * P;
}

template <int M, int T, bool FIRST_FROM_REGS, typename Real>
__device__ __forceinline__ void fft_lds(cx<Real>* buf, const cx<Real>* __restrict__ tw,
                                        cx<Real>* regs, int tid) {
    fft_passes<M, T, 0, FIRST_FROM_REGS, Real>(buf, tw, regs, tid);
}

// First-pass read order helper: the point index that regs[e] must hold when a
// caller feeds fft_lds<.., FIRST_FROM_REGS=true>.
template <int M, int T>
__device__ __forceinline__ int first_pass_index(int tid, int e) {
    constexpr int R0 = Plan<M, M / T>::radix(0);
    constexpr int NB = M / R0;
    const int b = e / R0, r = e % R0;
    return tid + b * T + r * NB;
}

}  // namespace mpx
