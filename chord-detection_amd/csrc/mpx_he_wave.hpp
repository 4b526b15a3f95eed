// Harmonic-Energy chroma for 4096-sample frames, fp64: ONE WAVE PER FRAME, the frame held in registers.
//
// The workgroup-per-frame kernel (mpx_he.hip) keeps the 2048-point complex FFT of a frame in 32 KB of LDS and walks
// it four times (radix 8.8.8.4, 8 points per thread): its time is LDS round trips + fp64 issue + three workgroup
// barriers per frame, at 4 waves per SIMD.  Here a wave64 owns a frame: 32 complex points per lane live in 128 VGPRs
// and the 4096-point real transform is two 1024-point complex ones, 1024 = 32 x 32, one per lane parity:
//   x[4m + j], j < 4, are four real sequences; even lanes transform zA[m] = x[4m] + i x[4m+1], odd lanes
//   zB[m] = x[4m+2] + i x[4m+3] (lane L holds the sample pairs 2(L + 64 n1), n1 < 32: coalesced loads);
//   A. a 32-point DFT over n1 IN REGISTERS (no exchange at all);
//   B. the twiddles W_1024^(l k1), l = L >> 1, are folded into the second transform (a modulated DFT, see hw_fft32_modulated);
//   C. one transpose through LDS inside each parity class (real parts, then imaginary parts, through the same 16.6 KB
//      per wave) hands lane (k1, parity) its whole row, and a second 32-point DFT in registers finishes ZA, ZB;
//   D. only for the bins the 48 windows look at: Y_j from the conjugate-symmetric split of ZA, ZB, and
//      X[k] = (Y0 + W^k Y1) + W^2k (Y2 + W^k Y3), W = W_4096 -- the last two radix-2 levels of the real transform are
//      never computed for the other 1800 bins.
// One LDS round trip instead of four, no workgroup barrier in the frame loop, a wave never waits for another wave.
// The price is 2 waves per SIMD (<= 256 VGPRs); the long register-resident runs have the instruction-level
// parallelism to cover fp64 latency without more waves.
//
// HALVES = 2 (round 5): the reference's own default frame, 8192 samples (harmonic_energy.py:14-16).  x[8m + r], r < 8, are
// eight real sequences with 1024-point transforms T_r, X[k] = sum_r W_8192^(r k) T_r[k]: the wave runs the pipeline above
// TWICE per frame, pass h on zA[m] = x[8m + 2h] + i x[8m + 2h + 1] and zB[m] = x[8m + 4 + 2h] + i x[8m + 5 + 2h] (sample
// pairs 4 (L + 64 n1) + 2h: the same coalesced 8-byte loads), and gets
//   X_h[k] = (T_2h + W^k T_2h+1) + W^4k (T_4+2h + W^k T_5+2h),   W = W_8192,       X[k] = X_0[k] + W^2k X_1[k]:
// phase D with W_8192^k where the 4096-sample frame has W_4096^k (the group factor is its FOURTH power), pass 0's X_0 at
// the window bins parked (4 KB per wave, through global memory: registers and LDS are full) and folded in at the end of
// pass 1.  The window table is 32 KB then ([pass][pair]; the mirror image of pass h is pass 1 - h), which leaves room for
// seven waves per CU.  Each pass uses 8 of every 16 bytes of the frame: fetched pass by pass, every cache line of the signal
// comes in twice, the second time seven microseconds later and no longer from the L2: 127 us per 8196 frames streamed from
// HBM against 90 with the input resident in L2 (scripts/dev/he8192_ab.py).  Three ways around it were built and measured in
// round 5, none kept: both passes' pairs fetched together into the registers the spectrum leaves (312 bytes per lane of
// scratch), the two passes unrolled into separate code (240), pass 1's pairs fetched at the start of pass 0 and parked in a
// global buffer of the wave's own (scratch-free with ONE load sequence over run-time strides, but the wait for those loads
// at the top of the pass costs more than the second fetch: 153 us streamed, 108 resident).
//
// Same arithmetic contract as he_kernel (harmonic_energy.py:42-67): x * hamming_sym(N) in fp64, real-split on the
// window bins only, |X|^2 maxima, fourth root of the 48 maxima, the reference's summation order inside a frame.
#pragma once
#include "mpx_fft.hpp"
#include "mpx_fft_dif.hpp"
#include "mpx_internal.hpp"

namespace mpx {

// XCD-aware bijective remap: workgroup b runs on XCD b%8 (observed dispatch
// order; a wrong guess only costs speed).  Give each XCD a contiguous block of
// frames so the (N-hop)-sample overlap between neighbours hits its own L2.
__device__ __forceinline__ long long xcd_contiguous(long long b, long long g) {
    const long long q = g >> 3, r = g & 7;
    const long long xcd = b & 7, slot = b >> 3;
    const long long base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

struct HeWaveArgs {
    const float* sig;
    long long n;
    const FrameDesc* desc;  // nullptr => frame f starts at f*hop
    long long num_frames;
    int hop;
    const double* whalf;      // [2048] hamming_sym(4096)[0..2047] (the other half is its mirror image)
    const cx<double>* tw;     // [2048] W_2048^j
    const int* wk0;           // windows as index ranges into the bin list
    const int* wk1;
    const double* ww;         // 1/harmonic per window
    const unsigned* slots;    // [nb][2] per window bin k, k' = k mod 1024: (ZA[k'] | ZA[-k'] << 16), (ZB[k'] | ZB[-k'] << 16): double
                              // indices into the bin-ordered LDS copy (hw_slot)
    const cx<double>* twnb;   // W_N^k per window bin (N = 4096, or 8192 with HALVES == 2)
    cx<double>* escratch;     // HALVES == 2: [gridDim.x * WAVES][64 * ROUNDS] pass 0's spectrum at the window bins
    int nb, nwin, wins_per_note, num_harmonic;
    int quad_tail;            // host: nwin == 48, 4 windows per note, 2 harmonics, every window 1..8 bins wide
    double* out;              // [F,12] per-frame chroma, never null: the sums over frames are taken over these rows
    double* partial;          // [gridDim.x,12] or null: sum of each workgroup's rows, in row order (a second, tiny launch adds them)
};

// Transpose buffer of a wave: the rows (k1, 0) and (k1, 1) interleaved element by element, 528 B per pair (512 + 16 of
// padding).  The writers of one register (lane L holds column L >> 1 of row (k1, L & 1)) then store 64 consecutive
// doubles, and the readers (lane 2 k1 + parity owns row (k1, parity), elements 16 B apart) start in 64 different 8-byte
// slots, 16 k1 + 8 parity modulo the bank sweep: no bank conflict however the hardware groups the lanes.  (The first
// layout, 264-byte rows per parity, collided two-way on the paired reads and on the stores: SQ_LDS_BANK_CONFLICT was
// 46 % of SQ_LDS_IDX_ACTIVE.)  The bin-ordered copy interleaves ZA and ZB the same way.
constexpr int HW_PAIR = 528;
constexpr int HW_XBUF = 32 * HW_PAIR;  // per wave: 16 896 B
__host__ __device__ constexpr int hw_slot(int parity, int k) { return 2 * (k & 1023) + parity; }

__host__ __device__ constexpr int hw_br5(int p) {
    return ((p & 1) << 4) | ((p & 2) << 2) | (p & 4) | ((p & 8) >> 2) | ((p & 16) >> 4);
}

// cos / sin of 2 pi e / 32, e < 16
__device__ constexpr double HW_C[16] = {1.0, 0.98078528040323044913, 0.92387953251128675613, 0.83146961230254523708,
                                         0.70710678118654752440, 0.55557023301960222474, 0.38268343236508977173, 0.19509032201612826785,
                                         0.0, -0.19509032201612826785, -0.38268343236508977173, -0.55557023301960222474,
                                         -0.70710678118654752440, -0.83146961230254523708, -0.92387953251128675613, -0.98078528040323044913};
__device__ constexpr double HW_S[16] = {0.0, 0.19509032201612826785, 0.38268343236508977173, 0.55557023301960222474,
                                         0.70710678118654752440, 0.83146961230254523708, 0.92387953251128675613, 0.98078528040323044913,
                                         1.0, 0.98078528040323044913, 0.92387953251128675613, 0.83146961230254523708,
                                         0.70710678118654752440, 0.55557023301960222474, 0.38268343236508977173, 0.19509032201612826785};

// d * W_32^E, E a compile-time exponent < 16
template <int E>
__device__ __forceinline__ cx<double> hw_mul_w32(cx<double> d) {
    if constexpr (E == 0) {
        return d;
    } else if constexpr (E == 8) {
        return {d.y, -d.x};
    } else if constexpr (E == 4) {
        return {HW_C[4] * (d.x + d.y), HW_C[4] * (d.y - d.x)};
    } else if constexpr (E == 12) {
        return {HW_C[4] * (d.y - d.x), -HW_C[4] * (d.x + d.y)};
    } else {
        constexpr double c = HW_C[E], s = HW_S[E];   // W = c - i s
        return {d.x * c + d.y * s, d.y * c - d.x * s};
    }
}

template <int S, int G, int J>
__device__ __forceinline__ void hw_bfly(cx<double>* r) {
    const cx<double> a = r[G + J], b = r[G + J + S];
    r[G + J] = cadd(a, b);
    r[G + J + S] = hw_mul_w32<J * (16 / S)>(csub(a, b));
}
template <int S, int G, int J>
__device__ __forceinline__ void hw_stage_j(cx<double>* r) {
    if constexpr (J < S) {
        hw_bfly<S, G, J>(r);
        hw_stage_j<S, G, J + 1>(r);
    }
}
template <int S, int G>
__device__ __forceinline__ void hw_stage_g(cx<double>* r) {
    if constexpr (G < 32) {
        hw_stage_j<S, G, 0>(r);
        hw_stage_g<S, G + 2 * S>(r);
    }
}
// 32-point DFT in registers, decimation in frequency: natural order in, r[p] = X[br5(p)] out
__device__ __forceinline__ void hw_fft32(cx<double>* r) {
    hw_stage_g<16, 0>(r);
    hw_stage_g<8, 0>(r);
    hw_stage_g<4, 0>(r);
    hw_stage_g<2, 0>(r);
    hw_stage_g<1, 0>(r);
}

// The same transform in its Cooley-Tukey form (natural order in, r[p] = X[br5(p)] out, like hw_fft32): the twiddle of a
// butterfly multiplies its SECOND INPUT, so it fuses into the additions -- u = a + W b in four FMAs, a - W b = 2a - u in two:
// six instructions where the Gentleman-Sande butterfly above takes eight ((a - b) first, then the product).  All butterflies
// of a group share one twiddle, exponent br(group) x span: 46 of the 80 are trivial (four instructions), 34 take six:
// 388 instructions against 456.
template <int E>
__device__ __forceinline__ void hw_bfly_ct(cx<double>& ra, cx<double>& rb) {
    const cx<double> a = ra, b = rb;
    if constexpr (E == 0) {
        ra = cadd(a, b);
        rb = csub(a, b);
    } else if constexpr (E == 8) {   // W = -i
        ra = {a.x + b.y, a.y - b.x};
        rb = {a.x - b.y, a.y + b.x};
    } else if constexpr (E == 4) {   // W = c (1 - i)
        const double s1 = b.x + b.y, s2 = b.y - b.x;
        ra = {fma(HW_C[4], s1, a.x), fma(HW_C[4], s2, a.y)};
        rb = {fma(-HW_C[4], s1, a.x), fma(-HW_C[4], s2, a.y)};
    } else if constexpr (E == 12) {  // W = -c (1 + i)
        const double s1 = b.y - b.x, s2 = b.x + b.y;
        ra = {fma(HW_C[4], s1, a.x), fma(-HW_C[4], s2, a.y)};
        rb = {fma(-HW_C[4], s1, a.x), fma(HW_C[4], s2, a.y)};
    } else {
        constexpr double c = HW_C[E], sn = HW_S[E];   // W = c - i sn
        cx<double> u;
        u.x = fma(c, b.x, fma(sn, b.y, a.x));
        u.y = fma(c, b.y, fma(-sn, b.x, a.y));
        ra = u;
        rb = {fma(2.0, a.x, -u.x), fma(2.0, a.y, -u.y)};
    }
}
__host__ __device__ constexpr int hw_brn(int v, int bits) {
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1) << (bits - 1 - i);
    return r;
}
template <int S, int G, int J>
__device__ __forceinline__ void hw_ct_j(cx<double>* r) {
    if constexpr (J < S) {
        // group index G / (2 S), 16 / S groups in the stage: exponent = bit-reversed group index x span
        constexpr int LG = S == 16 ? 0 : (S == 8 ? 1 : (S == 4 ? 2 : (S == 2 ? 3 : 4)));
        hw_bfly_ct<hw_brn(G / (2 * S), LG) * S>(r[G + J], r[G + J + S]);
        hw_ct_j<S, G, J + 1>(r);
    }
}
template <int S, int G>
__device__ __forceinline__ void hw_ct_g(cx<double>* r) {
    if constexpr (G < 32) {
        hw_ct_j<S, G, 0>(r);
        hw_ct_g<S, G + 2 * S>(r);
    }
}
// (the first level -- span 16, twiddle 1 -- is taken together with the window multiplication in the frame loop)
__device__ __forceinline__ void hw_fft32_ct_after_first_level(cx<double>* r) {
    hw_ct_g<8, 0>(r);
    hw_ct_g<4, 0>(r);
    hw_ct_g<2, 0>(r);
    hw_ct_g<1, 0>(r);
}
__device__ __forceinline__ void hw_fft32_ct(cx<double>* r) {
    hw_ct_g<16, 0>(r);
    hw_ct_g<8, 0>(r);
    hw_ct_g<4, 0>(r);
    hw_ct_g<2, 0>(r);
    hw_ct_g<1, 0>(r);
}

// The same transform of the MODULATED sequence x[c] * theta^c (theta per lane): in decimation in frequency the factor
// theta^c of a pair (c, c + S) splits into theta^c -- which stays with the half-length subsequence -- and theta^S on the
// second element, so every stage multiplies its second inputs by ONE per-lane constant th[s] = theta^S, fused into the
// butterfly: u = a + th b (four FMAs), a - th b = 2a - u (two).  Two instructions more per butterfly than the plain one,
// against four for a separate twiddle multiplication plus four for producing each twiddle.
template <int S, int G, int J>
__device__ __forceinline__ void hw_bfly_mod(cx<double>* r, cx<double> th) {
    const cx<double> a = r[G + J], b = r[G + J + S];
    cx<double> u;
    u.x = fma(th.x, b.x, fma(-th.y, b.y, a.x));
    u.y = fma(th.x, b.y, fma(th.y, b.x, a.y));
    const cx<double> d = {fma(2.0, a.x, -u.x), fma(2.0, a.y, -u.y)};
    r[G + J] = u;
    r[G + J + S] = hw_mul_w32<J * (16 / S)>(d);
}
template <int S, int G, int J>
__device__ __forceinline__ void hw_stage_mod_j(cx<double>* r, cx<double> th) {
    if constexpr (J < S) {
        hw_bfly_mod<S, G, J>(r, th);
        hw_stage_mod_j<S, G, J + 1>(r, th);
    }
}
template <int S, int G>
__device__ __forceinline__ void hw_stage_mod_g(cx<double>* r, cx<double> th) {
    if constexpr (G < 32) {
        hw_stage_mod_j<S, G, 0>(r, th);
        hw_stage_mod_g<S, G + 2 * S>(r, th);
    }
}
// th[0..4] = theta^16, theta^8, theta^4, theta^2, theta
__device__ __forceinline__ void hw_fft32_modulated(cx<double>* r, const cx<double>* th) {
    hw_stage_mod_g<16, 0>(r, th[0]);
    hw_stage_mod_g<8, 0>(r, th[1]);
    hw_stage_mod_g<4, 0>(r, th[2]);
    hw_stage_mod_g<2, 0>(r, th[3]);
    hw_stage_mod_g<1, 0>(r, th[4]);
}

// Phase boundary for the instruction scheduler: with 32 complex points per lane the frame loop is one enormous basic
// block, and instructions hoisted across phases (all 32 window reads before the first conversion ...) end in scratch --
// whose reloads wait on vmcnt, i.e. on the prefetch of the next frame.
__device__ __forceinline__ void hw_phase() { __builtin_amdgcn_sched_barrier(0); }
// opaque copy of the lane id: what is derived from it is rebuilt on the spot instead of living across the transforms
__device__ __forceinline__ int hw_opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// value of lane (l ^ X) of the same quad, X = 1 or 2
template <int X>
__device__ __forceinline__ double hw_quad_xor(double v) {
    constexpr int ctrl = X == 1 ? 0xB1 : 0x4E;   // quad_perm [1,0,3,2] / [2,3,0,1]
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), ctrl, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), ctrl, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// This lane's 32 sample pairs of one frame: pair n1 starts at sample 2*(lane + 64*n1)
// (HALVES == 2: of one PASS over an 8192-sample frame -- x points at sample 2h of the frame, pair n1 starts 4*(lane + 64*n1) on)
template <bool FAST, int HALVES = 1>
__device__ __forceinline__ void hw_load_frame(float2* raw, const float* __restrict__ x, int lane, int valid) {
    const float* p = x + 2 * HALVES * lane;
    if (FAST) {
#pragma unroll
        for (int e = 0; e < 32; ++e) raw[e] = *reinterpret_cast<const float2*>(p + 128 * HALVES * e);
    } else {
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const int s = HALVES * (2 * lane + 128 * e);   // (valid counts from x: the caller has taken 2h off)
            raw[e].x = s < valid ? p[128 * HALVES * e] : 0.f;
            raw[e].y = s + 1 < valid ? p[128 * HALVES * e + 1] : 0.f;
        }
    }
}

// shared LDS tables in front of the waves' buffers (bytes): whalf | slots | twnb | ww | wk | theta powers | frame counter
__host__ __device__ constexpr int hw_shared_bytes(int rounds, int nwin, int halves = 1) {
    return 16384 * halves + 8 * 64 * rounds + 16 * 64 * rounds + ((16 * nwin + 15) & ~15) + 5 * 32 * 16 + 16;
}

// FASTONLY: the host has checked that EVERY frame of the launch is whole, inside the signal and 8-byte aligned (one signal,
// even hop: the headline shape), so the kernel holds the branch-free loader alone.  The general instantiation carries the
// ragged loader as well, and the register allocation of a kernel is that of its worst path: with both, 24 bytes per lane
// of scratch are reserved (and set up per wave) that the fast path never touches.
// K2MASK, bit q: some window bin k (or its mirror 1024 - k) has (k mod 1024) >> 5 == q.  Only those rows of ZA / ZB go into
// the bin-ordered LDS copy: a store costs the LDS 6 cycles, and the 64 of this copy were a third of the kernel's LDS time.
// The reference's windows at 44.1 kHz (bins 45 ... 375 and their mirrors 649 ... 979) need 22 of the 32 rows: HW_K2_44K;
// the host launches that instantiation when the plan's own mask is a subset, the all-rows one otherwise.  (A run-time
// mask -- 64 scalar branches in the loop body -- cost 112 bytes per lane of scratch.)
constexpr unsigned HW_K2_ALL = 0xffffffffu, HW_K2_44K = 0x7ff00ffeu;
__host__ __device__ constexpr unsigned hw_k2_bits(int k) {
    return (1u << ((k & 1023) >> 5)) | (1u << (((1024 - (k & 1023)) & 1023) >> 5));
}
template <unsigned K2MASK, int P>
__device__ __forceinline__ void hw_store_rows(double* mine, const cx<double>* b, bool imag) {
    if constexpr (P < 32) {
        if constexpr ((K2MASK >> hw_br5(P)) & 1u) mine[64 * hw_br5(P)] = imag ? b[P].y : b[P].x;
        hw_store_rows<K2MASK, P + 1>(mine, b, imag);
    }
}

// PAIRED (8192-sample frames): TWO WAVES PER FRAME -- the waves 2j and 2j + 1 of a workgroup take the same frames, wave 2j
// pass 0 and wave 2j + 1 pass 1, at the same time: their loads hit the same cache lines within a fraction of a frame (one
// fetch from HBM serves both; with one wave running the passes one after the other every line came in twice, 127 against 90 us
// per 8196 frames).  Pass 0's spectrum at the window bins goes to its partner through 4 KB of LDS per pair, guarded by two
// LDS counters (written: frames published by pass 0 / consumed by pass 1); a wave that gets ahead of its partner waits there,
// so the two stay within one frame of each other.  Frames are dealt to the pairs round-robin (no LDS frame counter).
template <int WAVES, int ROUNDS, bool DEBUG, bool FASTONLY = false, unsigned K2MASK = HW_K2_ALL, int HALVES = 1, bool PAIRED = false>
__global__ __launch_bounds__(WAVES * 64, 1) void he_wave_kernel(HeWaveArgs a, cx<double>* dbg) {
    static_assert(HALVES == 1 || HALVES == 2, "4096- or 8192-sample frames");
    static_assert(!PAIRED || (HALVES == 2 && WAVES % 2 == 0), "pairs of waves: one per pass of an 8192-sample frame");
    constexpr int N = 4096 * HALVES, T = WAVES * 64, NBP = 64 * ROUNDS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* whalf = reinterpret_cast<double*>(smem);                            // 16 KB per pass, shared by the waves
    uint2* slots_lds = reinterpret_cast<uint2*>(smem + 16384 * HALVES);         // [NBP]
    cx<double>* twnb_lds = reinterpret_cast<cx<double>*>(slots_lds + NBP);       // [NBP]
    double* ww_lds = reinterpret_cast<double*>(twnb_lds + NBP);                 // [nwin]
    int* wk_lds = reinterpret_cast<int*>(ww_lds + a.nwin);                      // [2 nwin]
    unsigned* next_frame = reinterpret_cast<unsigned*>(smem + hw_shared_bytes(ROUNDS, a.nwin, HALVES) - 16);
    cx<double>* theta_lds = reinterpret_cast<cx<double>*>(smem + hw_shared_bytes(ROUNDS, a.nwin, HALVES) - 16 - 5 * 32 * 16);   // [5][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    char* xbuf = smem + hw_shared_bytes(ROUNDS, a.nwin, HALVES) + wave * HW_XBUF;
    // PAIRED: behind the waves' buffers, per pair [64 ROUNDS complex: pass 0's spectrum] and (at the very end) two counters
    // (addresses rebuilt from a scalar pair index where they are used: nothing of them lives in vector registers)
    auto pair_index = [&]() -> int { return __builtin_amdgcn_readfirstlane(hw_opaque(tid) >> 7); };
    auto pair_buf_of = [&](int pr) -> cx<double>* {
        return reinterpret_cast<cx<double>*>(smem + hw_shared_bytes(ROUNDS, a.nwin, HALVES) + WAVES * HW_XBUF) + pr * NBP;
    };
    auto pair_flag_of = [&](int pr) -> volatile unsigned* {
        return reinterpret_cast<volatile unsigned*>(smem + hw_shared_bytes(ROUNDS, a.nwin, HALVES) + WAVES * HW_XBUF + (WAVES / 2) * NBP * 16) + 2 * pr;
    };

    // Frames of this workgroup: a contiguous run (neighbouring frames share 3/4 of their samples), handed out to its
    // waves one at a time by an LDS counter.  The two waves of a SIMD do not run at the same speed (the hardware favours the
    // older one: 15 000 against 31 000 clocks per frame), equal shares leave the younger ones working alone at the end.
    // Which wave computed a frame changes nothing: every frame is a row of `out`, summed afterwards in a fixed order.
    const long long w = xcd_contiguous(blockIdx.x, gridDim.x);
    const long long per = (a.num_frames + gridDim.x - 1) / gridDim.x;
    const long long g0 = w * per;
    long long g1 = g0 + per;
    if (g1 > a.num_frames) g1 = a.num_frames;
    auto take = [&]() -> unsigned { return atomicAdd(next_frame, 1u); };   // one lane: the workgroup's next frame index
    auto grab = [&]() -> long long {   // wave-uniform
        unsigned t = 0;
        if (lane == 0) t = take();
        return g0 + (long long)__builtin_amdgcn_readfirstlane((int)t);
    };
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
    bool fast = g0 < g1 && (reinterpret_cast<uintptr_t>(a.sig) & 7) == 0;
    if (fast) {
        if (a.desc) {
            for (long long f = g0 + lane; f < g1; f += 64) fast = fast && a.desc[f].valid >= N && (a.desc[f].start & 1) == 0;
            fast = __all(fast);
        } else {
            fast = (a.hop & 1) == 0 && (g1 - 1) * (long long)a.hop + N <= a.n;
        }
    }
    if (tid == 0) *next_frame = 0;
    if (PAIRED && tid < WAVES) const_cast<unsigned*>(pair_flag_of(0))[tid] = 0;   // (wave 0's lanes: every pair's two counters)
    auto fill_tables = [&]() {
        for (int i = tid; i < 1024 * HALVES; i += T) reinterpret_cast<double2*>(whalf)[i] = reinterpret_cast<const double2*>(a.whalf)[i];
        for (int i = tid; i < NBP; i += T) {   // (entries past the last bin: bin 0, results dropped)
            slots_lds[i] = reinterpret_cast<const uint2*>(a.slots)[i < a.nb ? i : 0];
            twnb_lds[i] = a.twnb[i < a.nb ? i : 0];
        }
        // W_1024^(k1 * 16 >> s): the per-stage constants of the modulated second transform of row k1
        // (a.tw is W_(N/2)^j: W_2048 for the 4096-sample frame, W_4096 for the 8192-sample one)
        if (tid < 160) theta_lds[tid] = a.tw[(2 * HALVES * (tid & 31) * (16 >> (tid >> 5))) & (2048 * HALVES - 1)];
        for (int i = tid; i < a.nwin; i += T) {
            ww_lds[i] = a.ww[i];
            wk_lds[2 * i] = a.wk0[i];
            wk_lds[2 * i + 1] = a.wk1[i];
        }
    };
    __syncthreads();   // the counter

#ifdef HW_TRACE   // timing experiments only: shader-clock stamps of workgroup 0 at the phase boundaries, [wave][frame][16]
// (scheduling barriers on BOTH sides and a volatile s_memtime: without the first one the compiler let the read float up
//  over the phase's arithmetic and the per-phase figures were meaningless; frame totals were right)
#define HW_STAMP(pt)                                                                                                \
    do {                                                                                                            \
        hw_phase();                                                                                                 \
        if (dbg && blockIdx.x == 0 && nframe < 32) {                                                                \
            unsigned long long t_;                                                                                  \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : : "memory");                           \
            if (lane == 0) reinterpret_cast<long long*>(dbg)[(wave * 32 + nframe) * 16 + (pt)] = (long long)t_;     \
        }                                                                                                           \
        hw_phase();                                                                                                 \
    } while (0)
#else
#define HW_STAMP(pt) do {} while (0)
#endif
    const bool quad_tail = a.quad_tail != 0;
    auto run = [&](auto fast_tag, auto pass_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        constexpr int PASS = decltype(pass_tag)::value;   // PAIRED: this wave's pass, a compile-time constant of its loop
        float2 raw[32];
        // (PAIRED: the wave's number through readfirstlane -- as `tid >> 6` the frame index, the next one and the pending one were
        //  64-bit VECTOR values, eight registers of a kernel that is short of them: 20 bytes of scratch, reloaded in the frame loop)
        long long f = PAIRED ? g0 + (__builtin_amdgcn_readfirstlane(wave) >> 1) : grab();
        const long long f0 = f;
        (void)f0;
        if (f < g1) {   // the first frame is on its way while the tables are filled
            long long start;
            int valid;
            frame_span(f, start, valid);
            const int h0 = PAIRED ? PASS : 0;
            hw_load_frame<FAST, HALVES>(raw, a.sig + start + 2 * h0, lane, valid - 2 * h0);
        }
        fill_tables();
        __syncthreads();
        [[maybe_unused]] int nframe = 0;
        // The tail of a frame with the reference's window shape -- the maximum of eight |X|^2 already on their way from
        // LDS, two square roots, the quad sums, the store: a chain of dependent latencies with a dozen instructions in it --
        // is finished one iteration later, behind the first transform of the NEXT frame, which covers it.
        double pend_m = 0.0, pend_w = 0.0;
        long long pend_f = -1;
        auto finish_tail = [&]() {
            const int ol = hw_opaque(lane);
            // (PAIRED: the weight is read again here instead of travelling round the loop -- that kernel is two registers short)
            const double wgt = PAIRED ? ww_lds[ol < 48 ? ol : 47] : pend_w;
            const double t = sqrt(sqrt(pend_m)) * wgt;   // sqrt(|X|) of the maximum, times 1/harmonic
            // chroma = (max_0 + max_1/2) + (max_2 + max_3/2) over the quad: the reference's association (its sums start at 0.0)
            const double u = t + hw_quad_xor<1>(t);
            const double chroma = u + hw_quad_xor<2>(u);
            // bin n sits in lane 4n: pulled into lane n, so that the row is one 96-byte store of 12 neighbouring lanes
            const int src = (ol < 12 ? 4 * ol : ol) * 4;
            const double row = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(chroma)),
                                                __builtin_amdgcn_ds_bpermute(src, __double2loint(chroma)));
            if (ol < 12) a.out[pend_f * 12 + ol] = row;
        };
        // (one loop body for both passes, the pass a run-time value: with the two passes unrolled into separate code the
        //  register allocation of the kernel needs 240 bytes per lane of scratch)
        [[maybe_unused]] int h = PAIRED ? PASS : 0;   // HALVES == 2: the pass of frame f this iteration runs (PAIRED: this wave's, always)
        [[maybe_unused]] unsigned pair_seq = 0;   // PAIRED: frames this wave has finished
        while (f < g1) {
            cx<double> z[32];
            HW_STAMP(0);
            // this wave's next frame: asked for now, read once the window reads below have drained the LDS queue anyway
            unsigned grabbed = 0;
            const bool last_pass = PAIRED || HALVES == 1 || h == HALVES - 1;   // (uniform) the next iteration starts another frame
            if (!PAIRED && last_pass && hw_opaque(lane) == 0) grabbed = take();
#if defined(HW_FFT_A_GS) || defined(HW_WINDOW_SEPARATE)
            {
                // window pairs through LDS, eight at a time and one group ahead of their use (the scheduler, left alone,
                // requests each pair right before the multiplication and eats the LDS latency 32 times).  Pair index
                // m = lane + 64 n1 for n1 < 16, the mirrored pair 2047 - m (values swapped) above.
                const int ol = hw_opaque(lane);
                const char* wlo = reinterpret_cast<const char*>(whalf) + 16 * ol;
                const char* whi = reinterpret_cast<const char*>(whalf) + 16 * (63 - ol);
                auto wload = [&](int e) -> double2 {
                    return e < 16 ? *reinterpret_cast<const double2*>(wlo + 1024 * e)
                                  : *reinterpret_cast<const double2*>(whi + 1024 * (31 - e));
                };
                double2 wv[8], nx[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) wv[j] = wload(j);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (g < 3) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) nx[j] = wload(8 * (g + 1) + j);
                    }
                    hw_phase();
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int e = 8 * g + j;
                        const double w0 = e < 16 ? wv[j].x : wv[j].y, w1 = e < 16 ? wv[j].y : wv[j].x;
                        z[e] = {(double)raw[e].x * w0, (double)raw[e].y * w1};
                    }
                    hw_phase();
#pragma unroll
                    for (int j = 0; j < 8; ++j) wv[j] = nx[j];
                }
            }
#else
            {
                // Window and the FIRST butterfly level of transform A in one step.  That level pairs the points n1 = e and
                // e + 16 with twiddle 1: u = a wa + b wb, v = a wa - b wb -- one product and two FMAs per component where
                // window, then butterfly, took two products, a sum and a difference (32 instructions fewer per frame, and
                // b wb is not rounded on its own).  Window pairs through LDS, four points (e, e + 16) at a time and one group
                // ahead of their use (the scheduler, left alone, requests each pair right before its product and eats the LDS
                // latency 32 times): pair index m = lane + 64 n1 below 1024, the mirrored pair 2047 - m (values swapped) above.
                // (HALVES == 2: the table is [pass][pair]; the mirror image of pass h's upper half lies in pass 1 - h)
                const int ol = hw_opaque(lane);
                const char* wlo = reinterpret_cast<const char*>(whalf) + 16 * ol + (HALVES == 2 ? 16384 * h : 0);
                const char* whi = reinterpret_cast<const char*>(whalf) + 16 * (63 - ol) + (HALVES == 2 ? 16384 * (1 - h) : 0);
                auto wload = [&](int e) -> double2 {
                    return e < 16 ? *reinterpret_cast<const double2*>(wlo + 1024 * e)
                                  : *reinterpret_cast<const double2*>(whi + 1024 * (31 - e));
                };
                double2 wv[8], nx[8];   // [2 j]: point e = 4 g + j, [2 j + 1]: point e + 16
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    wv[2 * j] = wload(j);
                    wv[2 * j + 1] = wload(j + 16);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (g < 3) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            nx[2 * j] = wload(4 * (g + 1) + j);
                            nx[2 * j + 1] = wload(4 * (g + 1) + j + 16);
                        }
                    }
                    hw_phase();
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = 4 * g + j;
                        const double pax = (double)raw[e].x * wv[2 * j].x, pay = (double)raw[e].y * wv[2 * j].y;
                        const double bx = (double)raw[e + 16].x, by = (double)raw[e + 16].y;
                        const double wbx = wv[2 * j + 1].y, wby = wv[2 * j + 1].x;   // upper half: the mirrored pair, swapped
                        z[e] = {fma(bx, wbx, pax), fma(by, wby, pay)};
                        z[e + 16] = {fma(-bx, wbx, pax), fma(-by, wby, pay)};
                    }
                    hw_phase();
#pragma unroll
                    for (int j = 0; j < 8; ++j) wv[j] = nx[j];
                }
            }
#endif
            long long fn = PAIRED ? f + WAVES / 2 : (last_pass ? g0 + (long long)__builtin_amdgcn_readfirstlane((int)grabbed) : f);
            HW_STAMP(1);
            // A: DFT over n1 in registers; z[p] = A[k1 = br5(p)]
#if defined(HW_FFT_A_GS)
            hw_fft32(z);
#elif defined(HW_WINDOW_SEPARATE)
            hw_fft32_ct(z);
#else
            hw_fft32_ct_after_first_level(z);
#endif
            hw_phase();
            if (pend_f >= 0) {
                finish_tail();
                if constexpr (HALVES == 2) pend_f = -1;   // (a frame is two iterations here: once is enough)
            }
            hw_phase();
            HW_STAMP(2);
            // B: the twiddles W_1024^(column * k1) between the two transforms are not applied here: the reader of row k1 sees
            // them as a modulation theta^column, theta = W_1024^k1, of its input, and folds it into the second transform
            hw_phase();
            HW_STAMP(3);
            // C: transpose inside the parity class (real parts, then imaginary parts)
            cx<double> b[32], th[5];
            {
                const int ol = hw_opaque(lane);
                char* wr = xbuf + 8 * ol;
                const char* rd = xbuf + HW_PAIR * (ol >> 1) + 8 * (ol & 1);
#pragma unroll
                for (int st = 0; st < 5; ++st) th[st] = theta_lds[32 * st + (ol >> 1)];
#if defined(HW_ABL_NO_TRANSPOSE)   // ablation (results are garbage): where does a frame's time go?  tests/tools/he_wave_check.hip
#pragma unroll
                for (int c = 0; c < 32; ++c) b[c] = z[c];
                (void)wr;
                (void)rd;
#elif defined(HW_ABL_NO_TRANSPOSE_WRITES)
#pragma unroll
                for (int c = 0; c < 32; ++c) b[c].x = *reinterpret_cast<const double*>(rd + 16 * c) + z[c].x;
                wave_lds_fence();
#pragma unroll
                for (int c = 0; c < 32; ++c) b[c].y = *reinterpret_cast<const double*>(rd + 16 * c) + z[c].y;
                wave_lds_fence();
                (void)wr;
#elif defined(HW_ABL_NO_TRANSPOSE_READS)
#pragma unroll
                for (int p = 0; p < 32; ++p) *reinterpret_cast<double*>(wr + HW_PAIR * hw_br5(p)) = z[p].x;
                wave_lds_fence();
#pragma unroll
                for (int p = 0; p < 32; ++p) *reinterpret_cast<double*>(wr + HW_PAIR * hw_br5(p)) = z[p].y;
                wave_lds_fence();
#pragma unroll
                for (int c = 0; c < 32; ++c) b[c] = z[c];
                (void)rd;
#else
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
#endif
            }
            hw_phase();
            HW_STAMP(4);
            // second DFT over the 32 columns, modulated: b[p] = Z_parity[(lane >> 1) + 32 br5(p)], parity = lane & 1
            hw_fft32_modulated(b, th);
            hw_phase();
            HW_STAMP(5);
            const int ol = hw_opaque(lane);
            if constexpr (DEBUG) {
#pragma unroll
                for (int p = 0; p < 32; ++p) dbg[(f * 2 + (ol & 1)) * 1024 + (ol >> 1) + 32 * hw_br5(p)] = b[p];
            }
            // D: bin-ordered copy in LDS (real parts, then imaginary parts), from which the lanes of the window bins gather.
            // Their per-lane constants are fetched now (the 32 stores cover the latency) rather than held across the transforms.
            double* xb = reinterpret_cast<double*>(xbuf);
            double* mine = xb + ol;   // hw_slot(parity, (lane >> 1) + 32 q) = lane + 64 q
            cx<double> twk[ROUNDS];
            uint2 sl[ROUNDS];
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                sl[r] = slots_lds[ol + 64 * r];
                twk[r] = twnb_lds[ol + 64 * r];
            }
            int wstart = 0, wlast = 0;
            double wweight = 0.0;
            if (quad_tail) {
                const int wi = ol < 48 ? ol : 47;
                wstart = wk_lds[2 * wi];
                wlast = wk_lds[2 * wi + 1] - 1;
                wweight = ww_lds[wi];
            }
#ifdef HW_ABL_DPP_MODEL
            // COST MODEL (tests/tools/he_wave_check.hip only; results are garbage): what phase D would execute with a bin and its
            // mirror in neighbouring lanes (rows k1 and 32 - k1 of one quad) and the conjugate split by DPP -- no bin-ordered copy,
            // no gathers.  Per lane and row k2 = 1..11: the partner's mirror value (4 quad_perm moves), the fix for the two
            // self-mirrored rows of quad 0 (8 selects), E / D without the halves (4), W^k D from an LDS twiddle (4 + a read),
            // P (2), the pair's other parity by DPP (4 moves), W^2k (3), X (6), |X|^2 / 4 (3), a store of the result.
            double re[ROUNDS][4], mg[ROUNDS];
            (void)re;
            {
                const bool is0 = (ol >> 1) == 0, is16 = (ol >> 1) == 1;
                double accm = 0.0;
                {
                    long long start;
                    int valid;
                    frame_span(fn < g1 ? fn : f, start, valid);
                    hw_load_frame<FAST>(raw, a.sig + start, ol, valid);
                }
#pragma unroll
                for (int k2 = 1; k2 <= 11; ++k2) {
                    constexpr int dummy = 0;
                    (void)dummy;
                    const int p = hw_br5(k2), pm = 31 - p, p0 = k2 == 1 ? pm : hw_br5(32 - k2);
                    double mx = hw_quad_xor<2>(b[pm].x), my = hw_quad_xor<2>(b[pm].y);
                    mx = is16 ? b[pm].x : mx;
                    my = is16 ? b[pm].y : my;
                    mx = is0 ? b[p0].x : mx;
                    my = is0 ? b[p0].y : my;
                    const cx<double> E = {b[p].x + mx, b[p].y - my}, D = {b[p].x - mx, b[p].y + my};
                    const cx<double> w = twnb_lds[(ol >> 1) + 32 * (k2 & 7)];
                    const cx<double> P = cadd(E, mul_mi(cmul(w, D)));
                    const cx<double> Q = {hw_quad_xor<1>(P.x), hw_quad_xor<1>(P.y)};
                    const cx<double> X = cadd(P, cmul(cmul(w, w), Q));
                    const double m2 = 0.25 * (X.x * X.x + X.y * X.y);
                    xb[ol + 64 * k2] = m2;
                    accm += m2;
                }
#pragma unroll
                for (int r = 0; r < ROUNDS; ++r) mg[r] = accm;
            }
            wave_lds_fence();
#else
#ifndef HW_ABL_NO_BINCOPY
            hw_store_rows<K2MASK, 0>(mine, b, false);   // only the rows a window bin or its mirror lives in
#else
            mine[0] = b[0].x + b[7].x + b[13].y + b[31].x;
#endif
            wave_lds_fence();
            {
                // the registers of the real parts are free: request the next frame.  Unconditionally -- a wave without a
                // next frame re-reads its own, mostly from L2: under `if (fn < g1)` the compiler keeps both versions of
                // the 64 sample registers alive across the join (240 B of scratch, 93 instead of 44 us per launch)
                long long start;
                int valid;
                frame_span(fn < g1 ? fn : f, start, valid);
                if constexpr (HALVES == 1) {
                    hw_load_frame<FAST>(raw, a.sig + start, ol, valid);
                } else {   // the other pass of this frame, or pass 0 of the next one
                    const int hn = PAIRED ? h : (last_pass ? 0 : h + 1);
                    hw_load_frame<FAST, HALVES>(raw, a.sig + start + 2 * hn, ol, valid - 2 * hn);
                }
            }
            // pass 1: what pass 0 parked, asked for once the spectrum's registers are free (behind the next pass's samples in
            // the memory queue: they were requested above)
            [[maybe_unused]] cx<double> e_parked[HALVES == 2 ? ROUNDS : 1];
            [[maybe_unused]] cx<double>* mine_e = nullptr;
            if constexpr (HALVES == 2 && !PAIRED) {   // scalar base + a 32-bit lane offset rebuilt on the spot: nothing lives across the transforms
                const unsigned wv = (unsigned)__builtin_amdgcn_readfirstlane(hw_opaque(tid) >> 6);
                mine_e = a.escratch + (size_t)(blockIdx.x * (unsigned)WAVES + wv) * NBP + (unsigned)ol;
            }
            if constexpr (PAIRED) mine_e = pair_buf_of(pair_index()) + ol;   // (LDS)
            double re[ROUNDS][4], mg[ROUNDS];
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                re[r][0] = xb[sl[r].x & 0xffffu];
                re[r][1] = xb[sl[r].x >> 16];
                re[r][2] = xb[sl[r].y & 0xffffu];
                re[r][3] = xb[sl[r].y >> 16];
            }
            wave_lds_fence();
#ifndef HW_ABL_NO_BINCOPY
            hw_store_rows<K2MASK, 0>(mine, b, true);
#else
            mine[64] = b[1].y + b[9].x + b[21].y + b[30].y;
#endif
            wave_lds_fence();
            hw_phase();
            if constexpr (HALVES == 2 && !PAIRED) {
#pragma unroll
                for (int r = 0; r < ROUNDS; ++r)
                    e_parked[r] = mine_e[64 * r];
            }
            if constexpr (PAIRED) {
                // pass 0 may overwrite the pair's buffer once its partner has taken the frame before; pass 1 needs this
                // frame's to be there.  (LDS executes a wave's instructions in order: data, then counter.)
                const unsigned want = h == 0 ? pair_seq : pair_seq + 1;
                volatile unsigned* pair_flag = pair_flag_of(pair_index());
                volatile unsigned* flag = pair_flag + (h == 0 ? 1 : 0);   // [0] published by pass 0, [1] consumed by pass 1
                while (*flag < want) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                if (h == 1) {
#pragma unroll
                    for (int r = 0; r < ROUNDS; ++r) e_parked[r] = mine_e[64 * r];
                    wave_lds_fence();
                    if (hw_opaque(lane) == 0) pair_flag[1] = pair_seq + 1;
                }
            }
            HW_STAMP(6);
            // Y0 + W^k Y1 = E - i W^k D with E = (Z[k'] + conj Z[-k']) / 2, D = (Z[k'] - conj Z[-k']) / 2 (ZA), likewise
            // Y2 + W^k Y3 (ZB); X[k] = (Y0 + W^k Y1) + W^2k (Y2 + W^k Y3).  Lanes past the last bin redo bin 0.
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const cx<double> A = {re[r][0], xb[sl[r].x & 0xffffu]}, Am = {re[r][1], -xb[sl[r].x >> 16]};
                const cx<double> B = {re[r][2], xb[sl[r].y & 0xffffu]}, Bm = {re[r][3], -xb[sl[r].y >> 16]};
                const cx<double> EA = {0.5 * (A.x + Am.x), 0.5 * (A.y + Am.y)}, DA = {0.5 * (A.x - Am.x), 0.5 * (A.y - Am.y)};
                const cx<double> EB = {0.5 * (B.x + Bm.x), 0.5 * (B.y + Bm.y)}, DB = {0.5 * (B.x - Bm.x), 0.5 * (B.y - Bm.y)};
                const cx<double> PA = cadd(EA, mul_mi(cmul(twk[r], DA)));
                const cx<double> PB = cadd(EB, mul_mi(cmul(twk[r], DB)));
                const cx<double> tw2 = cmul(twk[r], twk[r]);
                cx<double> X;
                if constexpr (HALVES == 1) {
                    X = cadd(PA, cmul(tw2, PB));
                } else {
                    // this pass's four sequences: the second group is four samples on (W^4k); pass 0 parks its sum, pass 1
                    // adds its own, two samples on (W^2k)
                    X = cadd(PA, cmul(cmul(tw2, tw2), PB));
                    if (h == 0) mine_e[64 * r] = X;
                    else X = cadd(e_parked[r], cmul(tw2, X));
                }
                mg[r] = X.x * X.x + X.y * X.y;
            }
            wave_lds_fence();
#endif   // HW_ABL_DPP_MODEL
            hw_phase();
            HW_STAMP(7);
            if constexpr (PAIRED) {
                ++pair_seq;
                if (h == 0) {   // (uniform) publish: the stores above are in the LDS queue ahead of the counter
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (hw_opaque(lane) == 0) pair_flag_of(pair_index())[0] = pair_seq;
                    f = fn;
                    continue;
                }
            } else if constexpr (HALVES == 2) {
                if (h == 0) {   // (uniform) the frame's other pass comes next; nothing to reduce yet
                    h = 1;
                    continue;
                }
                h = 0;
            }
            double* mag = xb;              // |X|^2 per window bin, over the dead spectrum
            double* winmax = xb + NBP;
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const int i = ol + 64 * r;
                if (i < a.nb) mag[i] = mg[r];
            }
            wave_lds_fence();
            if (quad_tail) {
                // The reference's own window shape: lane wi < 48 owns window wi and the four windows of a note sit in one
                // quad.  Eight reads at once (indices clamped to the window's last bin: seeing a bin twice does not change
                // a maximum, harmonic_energy.py:58-62); the rest is finish_tail, one iteration later.
                double v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = mag[wstart + j < wlast ? wstart + j : wlast];
                double m = v[0];
#pragma unroll
                for (int j = 1; j < 8; ++j) m = v[j] > m ? v[j] : m;   // in bin order, like the reference's loop
                pend_m = m;
                if constexpr (!PAIRED) pend_w = wweight;
                pend_f = f;
                wave_lds_fence();
                HW_STAMP(9);
                f = fn;
                ++nframe;
                continue;
            }
            for (int wi = ol; wi < a.nwin; wi += 64) {  // half-open window maxima (harmonic_energy.py:58-62)
                double m = -INFINITY;
                const int k1 = wk_lds[2 * wi + 1], kl = k1 - 1;
                for (int k = wk_lds[2 * wi]; k < k1; k += 4) {
                    const double v0 = mag[k], v1 = mag[k + 1 < kl ? k + 1 : kl], v2 = mag[k + 2 < kl ? k + 2 : kl],
                                 v3 = mag[k + 3 < kl ? k + 3 : kl];
                    m = v0 > m ? v0 : m;
                    m = v1 > m ? v1 : m;
                    m = v2 > m ? v2 : m;
                    m = v3 > m ? v3 : m;
                }
                winmax[wi] = m < 0.0 ? m : sqrt(sqrt(m));   // sqrt(|X|) of the maximum; an empty window keeps -inf
            }
            wave_lds_fence();
            if (ol < 12) {   // chroma[n] = sum_octave ( sum_harmonic max/h ), same association as the reference
                double chroma = 0.0;
                const int base = ol * a.wins_per_note;
                for (int oc = 0; oc < a.wins_per_note; oc += a.num_harmonic) {
                    double note_sum = 0.0;
                    for (int hh = 0; hh < a.num_harmonic; ++hh) note_sum += winmax[base + oc + hh] * ww_lds[base + oc + hh];
                    chroma += note_sum;
                }
                a.out[f * 12 + ol] = chroma;
            }
            wave_lds_fence();
            HW_STAMP(9);
            f = fn;
            ++nframe;
        }
        if (pend_f >= 0) finish_tail();
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    if constexpr (PAIRED) {   // (two loops, one per pass: what a pass keeps alive does not burden the other's registers)
        const bool odd = __builtin_amdgcn_readfirstlane(wave & 1) != 0;
        if constexpr (FASTONLY) {
            if (odd) run(std::true_type{}, P1{});
            else run(std::true_type{}, P0{});
        } else if (fast) {
            if (odd) run(std::true_type{}, P1{});
            else run(std::true_type{}, P0{});
        } else {
            if (odd) run(std::false_type{}, P1{});
            else run(std::false_type{}, P0{});
        }
    } else if constexpr (FASTONLY) {
        run(std::true_type{}, P0{});
    } else {
        if (fast)
            run(std::true_type{}, P0{});
        else
            run(std::false_type{}, P0{});
    }

    // Which wave computed a frame is decided at run time, so the sum over frames is taken over the rows, in row order:
    // 16 strided sub-sums per bin, combined in a fixed order.  (The rows were stored by this workgroup: the barrier's
    // workgroup-scope release/acquire makes them visible.)  One row per workgroup; the library's sum_all_kernel adds those.
    if (a.partial) {
        __syncthreads();
        double* sh = reinterpret_cast<double*>(smem);
        const int bin = tid % 12, sub = tid / 12;
        if (sub < 16) {
            double t = 0.0;
            for (long long f = g0 + sub; f < g1; f += 16) t += a.out[f * 12 + bin];
            sh[sub * 12 + bin] = t;
        }
        __syncthreads();
        if (tid < 12) {
            double t = 0.0;
            for (int j = 0; j < 16; ++j) t += sh[j * 12 + tid];
            a.partial[w * 12 + tid] = t;
        }
    }

}

}  // namespace mpx
