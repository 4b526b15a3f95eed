// In-place decimation-in-frequency FFT in LDS with WAVE-LOCAL passes (gfx950, wave64).
//
// M complex points, T = M/8 threads, every thread owns 8 points per pass
// (radix 8, last pass radix 8/4/2).  Pass 0 combines points M/8 apart, i.e.
// across the whole workgroup; after it the transform has split into blocks of
// M/8 <= 512 points, and because a wave64 x 8 points = 512 points, every later
// pass of wave w touches only LDS positions [512 w, 512 w + 512): its inputs
// were written by the same wave.  So the whole FFT needs ONE workgroup barrier
// (after pass 0); the remaining passes are ordered by the in-order LDS pipeline
// of the wave itself and the four waves of a workgroup drift apart freely.
// Passes are in place (a butterfly stores where it loaded), so one address
// computation serves the load and the store of a pass, and no "everyone has
// read" barrier is needed either.  Output order is digit-reversed; callers
// index it with dif_pos<M>(k).
//
// Bank conflicts: position p lives in slot sigma(p) = p ^ c4*[bit4] ^ c5*[bit5]
// ^ c6*[bit6] (XOR swizzle of the low four slot bits).  The coefficients were
// found with scripts/lds_swizzle_search.py, which simulates the ds_read_b128
// (4 x 16 non-contiguous lanes, 64 banks) and ds_write_b128 (8 x 8 lanes, 32
// banks) lane groups of MI355X: with them every load and every store of every
// pass is conflict free, except the stores of the LAST pass (2-way) -- which
// the harmonic-energy kernel prunes to the few bins it needs anyway.
// sigma is linear over XOR, so sigma(base + r*S) = sigma(base) ^ sigma(r*S): one
// v_xor with a literal per access.
#pragma once
#include "mpx_fft.hpp"

namespace mpx {

template <int M> struct Swz;
template <> struct Swz<512>  { static constexpr int c4 = 1, c5 = 2, c6 = 12; };
template <> struct Swz<1024> { static constexpr int c4 = 2, c5 = 5, c6 = 8; };
template <> struct Swz<2048> { static constexpr int c4 = 1, c5 = 6, c6 = 8; };
template <> struct Swz<4096> { static constexpr int c4 = 1, c5 = 2, c6 = 12; };

template <int M>
__host__ __device__ __forceinline__ constexpr int sigma(int p) {
    return p ^ ((p & 16) ? Swz<M>::c4 : 0) ^ ((p & 32) ? Swz<M>::c5 : 0) ^ ((p & 64) ? Swz<M>::c6 : 0);
}

// (sigma(x) ^ x) must not touch the bits of x itself, or "(sb ^ K) + x" would differ from "sb ^ K ^ x".
template <int M>
__host__ __device__ __forceinline__ constexpr bool swz_disjoint(int x) { return ((sigma<M>(x) ^ x) & x) == 0; }
template <int M>
__device__ __forceinline__ void constexpr_assert_disjoint(int) {}
static_assert(swz_disjoint<2048>(32) && swz_disjoint<2048>(7 * 32) && swz_disjoint<2048>(7 * 4) && swz_disjoint<4096>(7 * 8) &&
              swz_disjoint<4096>(7 * 64) && swz_disjoint<1024>(7 * 16) && swz_disjoint<1024>(7 * 2) &&
              swz_disjoint<512>(7 * 64) && swz_disjoint<512>(7 * 8), "swizzle constants overlap a stride field");

// Slot of position base + r*S given sb = sigma(base) (base has no bits in r's field).  sigma only rewrites
// the low four bits, so for S >= 16 the stride part is a plain add (-> DS immediate offset) and only the
// swizzle part K_r = sigma(r*S) ^ (r*S) < 16 needs a v_xor; for S < 16 the stride field lies inside the
// swizzled bits and the whole thing is one xor.
template <int M, int S>
__device__ __forceinline__ int dif_addr(int sb, int r) {
    if constexpr (S >= 16)
        return (sb ^ (sigma<M>(r * S) ^ (r * S))) + r * S;
    else
        return sb ^ sigma<M>(r * S);
}

// Radices of the in-place plan: 8,8,8 then 8 / 4 / 2 / nothing.
template <int M>
struct DifPlan {
    static constexpr int LM = ilog2(M);
    static constexpr int full = LM / 3, rem = LM % 3;
    static constexpr int n = full + (rem ? 1 : 0);
    // (written without recursion: a recursive constexpr called with a loop variable is emitted as a real
    //  device FUNCTION CALL -- ABI spills and scratch -- instead of being folded after unrolling)
    __host__ __device__ __forceinline__ static constexpr int radix(int i) { return i < full ? 8 : (1 << rem); }
    __host__ __device__ __forceinline__ static constexpr int block(int i) {  // L of pass i
        return i <= full ? (M >> (3 * i)) : (M >> (3 * full + rem));
    }
    __host__ __device__ __forceinline__ static constexpr int stride(int i) { return block(i) / radix(i); }  // S of pass i
};

// LDS position (before sigma) of frequency k after the last pass (digit reversal).
template <int M>
__host__ __device__ __forceinline__ int dif_pos(int k) {
    using PL = DifPlan<M>;
    int pos = 0, remk = k;
#pragma unroll
    for (int i = 0; i < PL::n; ++i) {
        const int R = PL::radix(i);
        pos += (remk & (R - 1)) * PL::stride(i);
        remk >>= ilog2(R);
    }
    return pos;
}

// Frequency held at LDS position pos after the last pass (inverse of dif_pos).
template <int M>
__host__ __device__ __forceinline__ int dif_freq(int pos) {
    using PL = DifPlan<M>;
    int k = 0, mul = 1;
#pragma unroll
    for (int i = 0; i < PL::n; ++i) {
        const int R = PL::radix(i);
        k += ((pos / PL::stride(i)) & (R - 1)) * mul;
        mul *= R;
    }
    return k;
}

// One radix-R DIF butterfly of pass I for butterfly id b.  LOAD: take the inputs from LDS (else
// from v); STORE: put the outputs back in place (else leave them in v: v[q] = position base+q*S).
// Per-thread twiddle bases W_L^j of the twiddled passes (all but the last); loaded once, outside
// the caller's frame loop.
template <int M, typename Real>
struct DifTwiddles {
    cx<Real> w[DifPlan<M>::n - 1];
};

template <int M, typename Real>
__device__ __forceinline__ DifTwiddles<M, Real> dif_load_twiddles(const cx<Real>* __restrict__ tw, int tid) {
    using PL = DifPlan<M>;
    DifTwiddles<M, Real> t;
#pragma unroll
    for (int i = 0; i < PL::n - 1; ++i) {
        const int S = PL::stride(i), L = PL::block(i);
        t.w[i] = tw[(tid & (S - 1)) * (M / L)];
    }
    return t;
}

// `b` should be derived from an OPAQUE copy of the thread id when the caller loops over frames:
// every address below depends only on the thread, and a compiler that hoists them all out of the
// loop (8 per pass, plus store predicates) runs out of registers and spills.
template <int M, int I, bool LOAD, bool STORE, typename Real>
__device__ __forceinline__ int dif_butterfly(cx<Real>* buf, const DifTwiddles<M, Real>& twd, cx<Real>* v,
                                             int b) {
    using PL = DifPlan<M>;
    constexpr int R = PL::radix(I), L = PL::block(I), S = PL::stride(I);
    const int blk = b / S, j = b & (S - 1);
    const int base = blk * L + j;
    const int sb = sigma<M>(base);
    // sigma(base + r*S) = (sigma(base) ^ K_r) + r*S with K_r = sigma(r*S) ^ (r*S) < 16: base has no bits in
    // r's field, so r*S becomes an immediate offset of the DS instruction and only the few distinct K_r
    // need a v_xor each.
    if (LOAD) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            v[r] = buf[dif_addr<M, S>(sb, r)];
        }
    }
    SmallDft<R, Real>::run(v);
    if (S > 1) {
        // output q of a DIF butterfly is multiplied by W_L^(j q) = W_M^(j q M/L)
        cx<Real> w[R];
        w[1] = twd.w[I < PL::n - 1 ? I : 0];
        asm volatile("" : "+v"(w[1].x), "+v"(w[1].y));  // keep the power chain inside the frame loop (see mpx_fft.hpp)
#pragma unroll
        for (int q = 2; q < R; ++q) {
            const int hb = 1 << (31 - __builtin_clz(q));
            w[q] = (q == hb) ? cmul(w[q / 2], w[q / 2]) : cmul(w[hb], w[q - hb]);
        }
#pragma unroll
        for (int q = 1; q < R; ++q) v[q] = cmul(v[q], w[q]);
    }
    if (STORE) {
#pragma unroll
        for (int q = 0; q < R; ++q) buf[dif_addr<M, S>(sb, q)] = v[q];
    }
    return base;
}

// Butterfly id of thread `tid` in pass I (h-th butterfly of the thread; h > 0 only when the last
// radix is below 8).  The last pass uses a wave-local assignment.
template <int M, int I>
__host__ __device__ __forceinline__ int dif_bid(int tid, int h) {
    using PL = DifPlan<M>;
    constexpr int R = PL::radix(I);
    if (R == 8) return tid;
    constexpr int PER_WAVE = 512 / R;  // butterflies inside one wave's 512-point region
    return (tid >> 6) * PER_WAVE + (tid & 63) + 64 * h;
}

// Order LDS traffic of one wave across passes: the hardware executes a wave's DS instructions in
// order; this only stops the compiler from moving them.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Middle passes 1 .. n-2 (radix 8, one butterfly per thread, wave-local).
template <int M, int I, typename Real>
__device__ __forceinline__ void dif_middle(cx<Real>* buf, const DifTwiddles<M, Real>& twd, cx<Real>* v, int tid) {
    using PL = DifPlan<M>;
    if constexpr (I < PL::n - 1) {
        dif_butterfly<M, I, true, true, Real>(buf, twd, v, tid);
        wave_lds_fence();
        dif_middle<M, I + 1, Real>(buf, twd, v, tid);
    }
}

// Full transform.  In: regs[r] = point tid + r*T (r < 8).  Out: regs holds the outputs of this
// thread's last-pass butterflies, NOT stored: regs[h*R + q] is LDS position
// dif_last_pos(tid, h, q), frequency dif_freq(position).  One __syncthreads() inside.
template <int M, typename Real>
__device__ __forceinline__ void dif_fft_keep_last(cx<Real>* buf, const DifTwiddles<M, Real>& twd, cx<Real>* regs,
                                                  int tid) {
    using PL = DifPlan<M>;
    static_assert(PL::n >= 2 && PL::radix(0) == 8, "plan");
    dif_butterfly<M, 0, false, true, Real>(buf, twd, regs, tid);
    __syncthreads();
    dif_middle<M, 1, Real>(buf, twd, regs, tid);
    constexpr int RL = PL::radix(PL::n - 1);
#pragma unroll
    for (int h = 0; h < 8 / RL; ++h)
        dif_butterfly<M, PL::n - 1, true, false, Real>(buf, twd, regs + h * RL, dif_bid<M, PL::n - 1>(tid, h));
}

// The same transform with the caller's barrier moved INSIDE: the first pass's butterflies are computed in registers, then the
// workgroup barrier that protects `buf` from this transform's first stores (everybody has finished reading the previous
// contents), then the stores.  What a caller would otherwise wait for in front of the transform overlaps the butterfly arithmetic.
template <int M, typename Real>
__device__ __forceinline__ void dif_fft_keep_last_barrier_before_stores(cx<Real>* buf, const DifTwiddles<M, Real>& twd, cx<Real>* regs,
                                                                        int tid) {
    using PL = DifPlan<M>;
    static_assert(PL::n >= 2 && PL::radix(0) == 8, "plan");
    const int base = dif_butterfly<M, 0, false, false, Real>(buf, twd, regs, tid);
    __syncthreads();
    {
        constexpr int S = PL::stride(0);
        const int sb = sigma<M>(base);
#pragma unroll
        for (int q = 0; q < 8; ++q) buf[dif_addr<M, S>(sb, q)] = regs[q];
    }
    __syncthreads();
    dif_middle<M, 1, Real>(buf, twd, regs, tid);
    constexpr int RL = PL::radix(PL::n - 1);
#pragma unroll
    for (int h = 0; h < 8 / RL; ++h)
        dif_butterfly<M, PL::n - 1, true, false, Real>(buf, twd, regs + h * RL, dif_bid<M, PL::n - 1>(tid, h));
}

template <int M>
__host__ __device__ __forceinline__ int dif_last_pos(int tid, int h, int q) {
    using PL = DifPlan<M>;
    constexpr int RL = PL::radix(PL::n - 1);
    return dif_bid<M, PL::n - 1>(tid, h) * RL + q;
}

// ---------------------------------------------------------------- inverse (decimation in time)
// The transpose of the network above, run backwards: digit-reversed input, natural output, same LDS
// addresses pass by pass, so the wave-local property carries over -- the passes n-1 .. 1 stay inside
// a wave's 512-point region and only pass 0 (last here) needs the workgroup barrier.  Each butterfly
// undoes its forward twin: multiply by the CONJUGATE twiddles first, then the conjugate small DFT
// (re/im swap in and out of SmallDft).  Unscaled: idit(dif(x)) = M x.
template <typename Real>
__device__ __forceinline__ cx<Real> cmulc(cx<Real> a, cx<Real> w) {  // a * conj(w)
    return {a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y};
}

template <int M, int I, bool LOAD, bool STORE, typename Real>
__device__ __forceinline__ void idit_butterfly(cx<Real>* buf, const DifTwiddles<M, Real>& twd, cx<Real>* v, int b) {
    using PL = DifPlan<M>;
    constexpr int R = PL::radix(I), L = PL::block(I), S = PL::stride(I);
    const int blk = b / S, j = b & (S - 1);
    const int base = blk * L + j;
    const int sb = sigma<M>(base);
    if (LOAD) {
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = buf[dif_addr<M, S>(sb, r)];
    }
    if (S > 1) {
        cx<Real> w[R];
        w[1] = twd.w[I < PL::n - 1 ? I : 0];
        asm volatile("" : "+v"(w[1].x), "+v"(w[1].y));
#pragma unroll
        for (int q = 2; q < R; ++q) {
            const int hb = 1 << (31 - __builtin_clz(q));
            w[q] = (q == hb) ? cmul(w[q / 2], w[q / 2]) : cmul(w[hb], w[q - hb]);
        }
#pragma unroll
        for (int q = 1; q < R; ++q) v[q] = cmulc(v[q], w[q]);
    }
#pragma unroll
    for (int q = 0; q < R; ++q) v[q] = {v[q].y, v[q].x};
    SmallDft<R, Real>::run(v);
#pragma unroll
    for (int q = 0; q < R; ++q) v[q] = {v[q].y, v[q].x};
    if (STORE) {
#pragma unroll
        for (int q = 0; q < R; ++q) buf[dif_addr<M, S>(sb, q)] = v[q];
    }
}

template <int M, int I, typename Real>
__device__ __forceinline__ void idit_middle(cx<Real>* buf, const DifTwiddles<M, Real>& twd, cx<Real>* v, int tid) {
    if constexpr (I >= 1) {
        idit_butterfly<M, I, true, true, Real>(buf, twd, v, tid);
        wave_lds_fence();
        idit_middle<M, I - 1, Real>(buf, twd, v, tid);
    }
}

// Unscaled inverse transform.  In: regs in the layout dif_fft_keep_last leaves (regs[h*R + q] =
// spectrum value of LDS position dif_last_pos(tid, h, q), frequency dif_freq(position)).
// Out: regs[r] = point tid + r*T (T = M/8), not stored.  One __syncthreads() inside; the caller
// guarantees nobody else is still reading `buf` when a wave enters (only the wave's own 512-point
// region is written before that barrier).
template <int M, typename Real>
__device__ __forceinline__ void idit_fft_from_last(cx<Real>* buf, const DifTwiddles<M, Real>& twd, cx<Real>* regs,
                                                   int tid) {
    using PL = DifPlan<M>;
    constexpr int RL = PL::radix(PL::n - 1);
#pragma unroll
    for (int h = 0; h < 8 / RL; ++h)
        idit_butterfly<M, PL::n - 1, false, true, Real>(buf, twd, regs + h * RL, dif_bid<M, PL::n - 1>(tid, h));
    wave_lds_fence();
    idit_middle<M, PL::n - 2, Real>(buf, twd, regs, tid);
    __syncthreads();
    idit_butterfly<M, 0, true, false, Real>(buf, twd, regs, tid);
}

}  // namespace mpx
