// ESACF chroma (reference method 1) in fp64.  Four kernels per batch of frames:
//
//  1. bandsplit_kernel   one lane per frame: 12-stage warped all-pass cascade +
//                        13-tap warped FIR residual (dsp/wfir.py:25-43), then
//                        HP -> half-wave rectify -> LP and LP (esacf.py:47-51,
//                        132-134, dsp/lowpass.py:6-8); zero state per frame.
//  2. sacf_kernel        one workgroup per frame: N-point circular DFT of
//                        x_lo + i*x_hi (direct LDS FFT for power-of-two N,
//                        Bluestein chirp-z on the same LDS FFT otherwise),
//                        S = |X_lo|^0.67 + |X_hi|^0.67, DFT again, first
//                        (N-1)//2 lags (esacf.py:93-105); enhancement
//                        (esacf.py:108-129); peakutils.indexes peak picking
//                        (esacf.py:56).
//  3. peakfit_kernel     one thread per peak: gaussian LM fit (esacf.py:60).
//  4. scatter_kernel     one thread per frame: fs/tau -> pitch class -> 12 bins
//                        (esacf.py:64-71).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "mpx_fft_dif.hpp"
#include "mpx_internal.hpp"
#include "mpx_lm.hpp"
#include "mpx_pow067.hpp"

namespace mpx {

struct BandCoef {
    double a;        // bark warp coefficient
    double c[13];    // warped remez taps
    double lpb[3], lpa[3];
    double hpb[3], hpa[3];
};

// ------------------------------------------------------------------ kernel 1
// One lane per frame runs the sample-serial IIR chain; a wave owns 64 consecutive frames.  Input and the
// two outputs move through LDS tiles of 64 frames x 16 samples so that global traffic is row segments
// (64 B in, 128 B out per frame and tile) instead of one isolated element per lane: lanes 16r..16r+15 of
// a transfer instruction cover 16 consecutive samples of frame r.  Tile rows are padded to 17 elements:
// conflict free both for the per-lane row walk and for the row-segment transfers.
constexpr int BS_TILE = 16;

// Band-split output layout: kernel 1 writes whole LDS tiles (64 frames x 16 samples of (x_lo, x_hi) pairs,
// 16 KB) back to back -- streaming writes instead of 128-byte pieces 16 KB apart, which held the kernel at
// 1.8 TB/s -- and kernel 2 gathers a frame as 256-byte pieces.  Index of sample n of (batch-local) frame f:
__host__ __device__ inline size_t band_index(long long f, int n, int N) {
    const int ntiles = (N + BS_TILE - 1) / BS_TILE;
    return (((size_t)(f >> 6) * ntiles + (n >> 4)) * 64 + (size_t)(f & 63)) * BS_TILE + (n & 15);
}

// A frame CUT IN TIME (BandCut, chosen by the host from the frame length and the filters alone -- never from the batch, so
// a frame's bits do not depend on what it is batched with): gridDim.y pieces, piece 0 = tiles [0, p0), piece p >= 1 =
// tiles [p0 + (p - 1) pt, p0 + p pt); a piece other than the first starts `rt` tiles early from ZERO state and throws that
// run-in away.  Every stage is a stable linear filter (the rectifier between them has no memory), so what the true state
// at the piece's start would have added has decayed below 2^-60 of the filters' gain by the time the piece's own samples
// begin (band_cut: the run-in is sized from the simulated impulse responses).  Piece 0 is the reference's zero-state start.
struct BandCut {
    int p0, pt, rt;   // tiles of 16 samples; one piece: p0 = all tiles
};

template <bool XW>
__global__ __launch_bounds__(64) void bandsplit_kernel(const float* __restrict__ sig, long long n,
                                                       const FrameDesc* __restrict__ desc, long long frame0,
                                                       long long num_frames, int N, int hop, BandCoef k,
                                                       cx<double>* __restrict__ xb,
                                                       double* __restrict__ xw, BandCut cut) {
    __shared__ float tin[64][BS_TILE + 1];
    __shared__ double tlo[64][BS_TILE + 1];
    __shared__ double thi[64][BS_TILE + 1];
    __shared__ double twf[XW ? 64 : 1][BS_TILE + 1];
    __shared__ long long row_start[64];
    __shared__ int row_valid[64];
    const int lane = threadIdx.x;
    const long long lf0 = (long long)blockIdx.x * 64;  // first frame (within this batch) of the wave
    {
        const long long lf = lf0 + lane;
        long long start = 0;
        int valid = 0;
        if (lf < num_frames) {
            const long long f = frame0 + lf;
            if (desc) {
                start = desc[f].start;
                valid = desc[f].valid;
            } else {
                start = f * (long long)hop;
                const long long left = n - start;
                valid = left >= N ? N : (left > 0 ? (int)left : 0);
            }
        }
        row_start[lane] = start;
        row_valid[lane] = valid;
    }
    __syncthreads();
    // The chain of a sample is 12 all-pass stages deep plus the band filters: evaluated sample by sample it is ONE
    // dependent chain of ~30 fp64 operations, and a lone wave (8192 frames are 128 waves on 1024 SIMDs) pays the full
    // latency on each -- 516 clocks per sample for 240 clocks of issue.  Software pipelining across samples removes that
    // (the Iterative-F0 front end does the same): in iteration tau all-pass stage i works on sample tau - i, reading what
    // stage i - 1 produced one iteration earlier, and the band filters on sample tau - 12.  All 13 stage updates of an
    // iteration are then independent of each other (stages are walked in reverse, so an input is consumed before it is
    // overwritten).  Every stage performs the operations of the sequential form on the same operands; which product of
    // a sum of two products the compiler fuses is its choice here as it was there (results agree to an ulp).
    double z[12], pin[12], pxh[12];   // all-pass state; pipeline registers feeding all-pass i (its input, the partial x_hat)
#pragma unroll
    for (int i = 0; i < 12; ++i) z[i] = pin[i] = pxh[i] = 0.0;
    double h1 = 0, h2 = 0, g1 = 0, g2 = 0, l1 = 0, l2 = 0;
    double fxh = 0.0;                 // x_hat of the sample the band filters take next
    float xr[BS_TILE];                // the last 16 raw samples (a sample waits 12 iterations for its x_hat)
#pragma unroll
    for (int q = 0; q < BS_TILE; ++q) xr[q] = 0.f;
    // The workgroup is ONE wave: its LDS operations execute in program order, so the tiles need no
    // s_barrier -- and above all no __syncthreads(), whose release fence would make every tile wait for its
    // own 16 KB of global stores to retire.  The input of tile t+1 is fetched into registers while tile t runs.
    // (loads are unconditional, from a clamped address, and the zero padding is a select: with a conditional load the
    //  compiler branches per element and waits for each row-table read -- 32 LDS round trips per tile, half the time of
    //  a small batch)
    float nxt[BS_TILE];
    auto fetch = [&](int t0) {
        long long st[BS_TILE];
        int vl[BS_TILE];
#pragma unroll
        for (int q = 0; q < BS_TILE; ++q) {
            const int r = (q * 64 + lane) >> 4;
            st[q] = row_start[r];
            vl[q] = row_valid[r];
        }
#pragma unroll
        for (int q = 0; q < BS_TILE; ++q) {
            const int t = t0 + (lane & 15);
            const bool ok = t < vl[q];
            const float v = sig[ok ? st[q] + t : 0];
            nxt[q] = ok ? v : 0.f;
        }
    };
    const int ntiles = (N + BS_TILE - 1) / BS_TILE;
    // this wave's piece of the frames: output tiles [tile_lo, tile_hi), filtering from tile_in (zero state there)
    const int piece = blockIdx.y;
    const int tile_lo = piece == 0 ? 0 : cut.p0 + (piece - 1) * cut.pt;
    const int tile_hi = piece == 0 ? cut.p0 : (tile_lo + cut.pt < ntiles ? tile_lo + cut.pt : ntiles);
    const int tile_in = tile_lo - cut.rt > 0 && piece > 0 ? tile_lo - cut.rt : 0;
    // stage out: a tile goes out exactly as it lies, (x_lo, x_hi) pairs, 16 KB contiguous per wave:
    // band layout [block of 64 frames][tile of 16 samples][frame][sample] (band_index above)
    auto flush = [&](int tile) {
        cx<double>* dst = xb + ((size_t)blockIdx.x * ntiles + tile) * (64 * BS_TILE);
#pragma unroll
        for (int q = 0; q < BS_TILE; ++q) {
            const int e = q * 64 + lane, r = e >> 4, c = e & 15;
            dst[e] = {tlo[r][c], thi[r][c]};
            if (XW) {
                const int t = tile * BS_TILE + c;
                if (t < N && lf0 + r < num_frames) xw[(lf0 + r) * (long long)N + t] = twf[r][c];
            }
        }
    };
    fetch(tile_in * BS_TILE);
    // block tb runs iterations tau = tb .. tb + 15: it takes in the samples tb .. tb + 15 and puts out the samples
    // tb - 12 .. tb + 3, i.e. columns 4 .. 15 of output tile tb / 16 - 1 (complete after iteration tb + 11: flushed there)
    // and columns 0 .. 3 of the next one.  One block past the last input tile drains the pipeline (its input is zeros).
    for (int tb = tile_in * BS_TILE; tb <= tile_hi * BS_TILE; tb += BS_TILE) {
        // ---- stage in: 4 frame rows x 16 samples per instruction
#pragma unroll
        for (int q = 0; q < BS_TILE; ++q) {
            const int e = q * 64 + lane;
            tin[e >> 4][e & 15] = nxt[q];
        }
        wave_lds_fence();
        if (tb + BS_TILE < N) fetch(tb + BS_TILE);
        else {
#pragma unroll
            for (int q = 0; q < BS_TILE; ++q) nxt[q] = 0.f;
        }
#pragma unroll
        for (int q = 0; q < BS_TILE; ++q) {
            // ---- band filters (sample tau - 12): residual, high-pass / rectifier / low-pass, low-pass
            {
                const double xs = (double)xr[(q + 4) & 15];   // written 12 iterations ago
                const double r = xs - fxh;
                const double yh = k.hpb[0] * r + h1;
                h1 = (h2 + k.hpb[1] * r) - k.hpa[1] * yh;
                h2 = k.hpb[2] * r - k.hpa[2] * yh;
                const double rect = yh < 0.0 ? 0.0 : yh;
                const double yhl = k.lpb[0] * rect + g1;
                g1 = (g2 + k.lpb[1] * rect) - k.lpa[1] * yhl;
                g2 = k.lpb[2] * rect - k.lpa[2] * yhl;
                const double yl = k.lpb[0] * r + l1;
                l1 = (l2 + k.lpb[1] * r) - k.lpa[1] * yl;
                l2 = k.lpb[2] * r - k.lpa[2] * yl;
                tlo[lane][(q + 4) & 15] = yl;
                thi[lane][(q + 4) & 15] = yhl;
                if (XW) twf[lane][(q + 4) & 15] = r;
            }
            // ---- all-pass stages 11 .. 1 (samples tau - 11 .. tau - 1), dsp/wfir.py:25-43
            {
                const double o = -k.a * pin[11] + z[11];   // b0*x + z
                z[11] = pin[11] + k.a * o;                  // b1*x - a1*y with b1 = 1, a1 = -a
                fxh = pxh[11] + k.c[12] * o;
            }
#pragma unroll
            for (int i = 10; i >= 1; --i) {
                const double o = -k.a * pin[i] + z[i];
                z[i] = pin[i] + k.a * o;
                pin[i + 1] = o;
                pxh[i + 1] = pxh[i] + k.c[i + 1] * o;
            }
            // ---- all-pass stage 0 on the new sample tau
            {
                const float xf = tin[lane][q];
                const double xt = (double)xf;
                const double xhat0 = k.c[0] * xt;
                const double o = -k.a * xt + z[0];
                z[0] = xt + k.a * o;
                pin[1] = o;
                pxh[1] = xhat0 + k.c[1] * o;
                xr[q] = xf;
            }
            if (q == 11 && tb > tile_lo * BS_TILE) {   // output tile tb / 16 - 1 is complete (and this piece's to write)
                wave_lds_fence();
                flush(tb / BS_TILE - 1);
                wave_lds_fence();
            }
        }
        wave_lds_fence();
    }
}

// debug taps (mpx_esacf_stage XLO / XHI): band layout -> [F,N] rows
__global__ __launch_bounds__(256) void band_unpack_kernel(const cx<double>* __restrict__ xb, long long num_frames, int N,
                                                          int which, double* __restrict__ out) {
    const long long f = blockIdx.x;
    for (int n = threadIdx.x; n < N; n += 256) {
        const cx<double> v = xb[band_index(f, n, N)];
        out[f * (long long)N + n] = which ? v.y : v.x;
    }
}

// ------------------------------------------------------------------ kernel 2
struct SacfArgs {
    const cx<double>* xb;  // (x_lo, x_hi), band layout
    int N, Mh;
    const cx<double>* tw;     // W_L
    const cx<double>* chirp;  // b[n] = exp(i*pi*n^2/N), n < N   (Bluestein only)
    const cx<double>* bhat;   // FFT_L(chirp filter) / L         (Bluestein only)
    const cx<double>* twn;    // W_N^k, k <= N/2                 (sacf_split_kernel only; chirp / bhat are those of N/2 there)
    int n_peaks_elim;
    double peak_thresh;
    int peak_min_dist;
    int enhance_mode;
    int defer_enhance;  // 1: stop after the SACF (time_stretch is a real phase vocoder for this many lags)
    int maxp;
    double* sacf_out;   // optional [F,Mh]
    double* y_out;      // [F,Mh] enhanced SACF
    int* peak_count;    // [F]
    int* peak_idx;      // [F,maxp]
    int* total_peaks;   // [0] long fits queued, [1] next work item (fit kernel), [2] other fits queued  ([0], [2]: written by the fit kernel from the shards)
    int* worklist;      // packed (frame << 12 | slot).  WL_SHARDS regions of `shard_cap` items: frame f queues in region f % WL_SHARDS, long fits from its front, others from its back
    int worklist_cap;   // = WL_SHARDS * shard_cap
    int shard_cap;      // items per region: ceil(frames / WL_SHARDS) * maxp
    unsigned long long* shard_cnt;   // [WL_SHARDS], 128 bytes apart: long fits queued (low half) | other fits queued (high half)
    long long num_frames;  // frames in this launch
    int pair;              // 1: a workgroup takes two frames and shares the second DFT between them
    int ablate;         // profiling knob (env MPX_SACF_ABLATE): 1 no pow, 2 no peak picking, 4 no 2nd DFT, 8 no 1st DFT
    // prime-factor engine (sacf_pfa_kernel): N = A0 * 3 * 11 * 31
    const unsigned short* pfa_pos;    // [N]   array position of index n (input sample n; output lag n)
    const double* pow_tab;            // [193] tables of mpx_pow067.hpp (|X|^0.67)
    const unsigned short* pfa_pairs;  // [npairs][2] positions (p, mirror p), p <= mirror p
    const cx<double>* pfa_cs31;       // [16][16] (cos, sin)(2 pi n k / 31), k = 0..15, n = 0..15
    const cx<double>* pfa_cs11;       // [6][5]   (cos, sin)(2 pi n k / 11)
    int pfa_npairs;
};

__device__ __forceinline__ cx<double> cconj(cx<double> a) { return {a.x, -a.y}; }
__device__ __forceinline__ cx<double> cswap(cx<double> a) { return {a.y, a.x}; }

// N-point forward DFT on the in-place DIF / inverse-DIT LDS engine (mpx_fft_dif.hpp), T = L/8
// threads, 8 points per thread.  In: regs[r] = x[tid + r*T] (anything for indices >= N).
//   BLUE:  Bluestein chirp-z as a length-L circular convolution (L >= 2N-1): chirp multiply ->
//          DIF (spectrum stays in registers, digit-reversed) -> filter spectrum (stored in that same
//          register order) -> inverse DIT -> chirp multiply.  Out: regs[r] = X[tid + r*T], natural.
//          Two workgroup barriers in total (one per FFT).
//   !BLUE: N == L, plain DIF.  Out: regs[e] = X[dif_freq(dif_last_pos(tid, e / RL, e % RL))].
// Entry condition: no wave is still reading `buf`.
template <int L, bool BLUE>
__device__ __forceinline__ void dft_regs(cx<double>* buf, const DifTwiddles<L, double>& twd, const SacfArgs& a,
                                         cx<double>* regs, int tid, const cx<double>* chv) {
    constexpr int T = L / 8;
    if (BLUE) {
        const int N = a.N;
        // N <= L/2 = 4T: only registers r < 4 hold input or output bins, and both chirp multiplications of both
        // transforms of a frame use the same four chirp values chv[r] = chirp[tid + r*T] (loaded once per frame)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int n = tid + r * T;
            regs[r] = (r < 4 && n < N) ? cmulc(regs[r], chv[r & 3]) : cx<double>{0.0, 0.0};
        }
        dif_fft_keep_last<L, double>(buf, twd, regs, tid);
#pragma unroll
        for (int e = 0; e < 8; ++e) regs[e] = cmul(regs[e], a.bhat[e * T + tid]);
        idit_fft_from_last<L, double>(buf, twd, regs, tid);
#pragma unroll
        for (int r = 0; r < 4; ++r) regs[r] = cmulc(regs[r], chv[r]);
    } else {
        dif_fft_keep_last<L, double>(buf, twd, regs, tid);
    }
}

// a lane's view of another lane's double through DPP (register to register, ~8 cycles; a shuffle is two ds_bpermute
// round trips through the LDS crossbar with a wait)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

// workgroup max and min of one double per thread; result in every thread.  Within a wave: four DPP steps make the 16
// lanes of a row agree, four v_readlane join the rows (until round 5: twelve shuffles of two ds_bpermute each)
template <int T>
__device__ __forceinline__ void block_minmax(double* sh, double& mx, double& mn) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#define MPX_MINMAX_STEP(CTRL)                                        \
    {                                                                \
        const double a_ = dpp_f64<CTRL>(mx), b_ = dpp_f64<CTRL>(mn); \
        mx = a_ > mx ? a_ : mx;                                      \
        mn = b_ < mn ? b_ : mn;                                      \
    }
    MPX_MINMAX_STEP(0xB1)   // quad_perm [1,0,3,2]
    MPX_MINMAX_STEP(0x4E)   // quad_perm [2,3,0,1]
    MPX_MINMAX_STEP(0x141)  // row_half_mirror
    MPX_MINMAX_STEP(0x140)  // row_mirror
#undef MPX_MINMAX_STEP
#pragma unroll
    for (int r = 16; r < 64; r += 16) {
        const double a = readlane_f64(mx, r), b = readlane_f64(mn, r);
        mx = a > mx ? a : mx;
        mn = b < mn ? b : mn;
    }
    mx = readlane_f64(mx, 0);  // lanes of the rows 1..3 have not seen row 0's values the way lane 0 has
    mn = readlane_f64(mn, 0);
    if (lane == 0) {
        sh[wave] = mx;
        sh[T / 64 + wave] = mn;
    }
    __syncthreads();
    mx = sh[0];
    mn = sh[T / 64];
#pragma unroll
    for (int w = 1; w < T / 64; ++w) {
        mx = sh[w] > mx ? sh[w] : mx;
        mn = sh[T / 64 + w] < mn ? sh[T / 64 + w] : mn;
    }
    __syncthreads();
}

// peakutils.indexes(y, thres, min_dist) on the Mh values in yv (LDS) + publication of the kept peaks.
// `scratch` is >= peak_scratch_bytes(Mh) = peak_words(Mh)*32 + (Mh/2 + 2)*8 bytes of LDS that nobody else uses any more.
//
// Everything is flag words: position i = tid + e*T of wave w is bit `lane` of word e*T/64 + w, so a
// __ballot is one word.  "dy != 0" words give the nearest non-zero neighbour of a plateau position
// with a clz/ffs (plateau rule); candidate words and kept words give compaction ranks with popcounts;
// no scans, no serial loops.  min_dist suppression is a parallel fixed point of peakutils' greedy
// rule: a candidate is kept iff no HIGHER candidate within min_dist is kept (height, then larger
// index, decides "higher" -- the visiting order of peakutils), so a candidate can decide as soon as
// all its higher neighbours have; every round decides at least the highest undecided one.
// flag words per set: >= (Mh + T) / 64 for T <= 512; at most 64, one per lane of the wave that ranks them (Mh <= 4095)
// (above 4095 lags -- sacf_huge_kernel's rows -- up to 128 words, two per lane of the ranking wave: peak_pick<T, true>)
__host__ __device__ inline int peak_words(int Mh) { return Mh <= 2047 ? 40 : (Mh <= 4095 ? 64 : 128); }
__host__ __device__ inline size_t peak_scratch_bytes(int Mh) {
    return ((size_t)peak_words(Mh) * 32 + (size_t)(Mh / 2 + 2) * 8 + 15) & ~(size_t)15;
}

// exclusive prefix sums over up to 64 (WIDE: 128) per-word counts held as c0 = count(word lane), c1 = count(word lane + 64);
// word_base() reads the sum in front of word w (wave-uniform w)
// (inclusive sums by DPP: row_shr 1, 2, 4, 8 inside the rows of 16 lanes, row_bcast:15 / row_bcast:31 across them -- six
// register-to-register steps; until round 5 six __shfl_up, a ds_bpermute round trip each)
__device__ __forceinline__ int wave_inclusive_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1 (lanes without a source add 0)
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return v;
}
template <bool WIDE>
struct WordScan {
    int e0, e1, total;
    __device__ __forceinline__ WordScan(int c0, int c1, int lane) {
        const int incl = wave_inclusive_sum(c0);
        e0 = incl - c0;
        total = __builtin_amdgcn_readlane(incl, 63);
        e1 = 0;
        if (WIDE) {
            const int incl1 = wave_inclusive_sum(c1);
            e1 = total + incl1 - c1;
            total += __builtin_amdgcn_readlane(incl1, 63);
        }
    }
    // w is the same in every lane of the wave: a v_readlane
    __device__ __forceinline__ int word_base(int w) const {
        w = __builtin_amdgcn_readfirstlane(w);
        return (WIDE && w >= 64) ? __builtin_amdgcn_readlane(e1, w - 64) : __builtin_amdgcn_readlane(e0, w);
    }
};

// The minimum-distance rounds of peak_pick for a frame of at most 64 candidates (the usual frame has a few dozen), one
// candidate per lane of wave 0 and no LDS inside the rounds.  Candidates are in ascending position, so the ones within
// min_dist of lane c are lanes c - 1, c - 2, ... and c + 1, c + 2, ... up to the first one too far away: d wave shifts
// (DPP wave_shr:1 / wave_shl:1) bring the d-th neighbour's position and height, once, for the masks "left neighbour d
// is higher" / "right neighbour d is higher or equal" (ties go to the larger index); a round then shifts the states
// along and applies the rule of the LDS version below -- removed if a higher neighbour is kept, kept if none is
// undecided -- whose fixed point does not depend on the order of the updates.  Returns false (nothing written) when a
// candidate has more than 31 neighbours on a side in range: the LDS version takes the frame.
__device__ __forceinline__ bool peak_rounds_in_lanes(const int* cand, const double* yv, volatile int* state, int ncand, int md, int lane) {
    const bool on = lane < ncand;
    const int pos = on ? cand[lane] : 0x3fffffff;
    const double v = on ? yv[pos] : 0.0;
    int pl = pos, pr = pos, vl_lo = __double2loint(v), vl_hi = __double2hiint(v), vr_lo = vl_lo, vr_hi = vl_hi;
    unsigned hl = 0, hr = 0;
    int dmax = 0;
    for (int d = 1;; ++d) {
        pl = __builtin_amdgcn_update_dpp(-0x3fffffff, pl, 0x138, 0xf, 0xf, false);  // wave_shr:1: lane c sees lane c - d
        vl_lo = __builtin_amdgcn_update_dpp(0, vl_lo, 0x138, 0xf, 0xf, false);
        vl_hi = __builtin_amdgcn_update_dpp(0, vl_hi, 0x138, 0xf, 0xf, false);
        pr = __builtin_amdgcn_update_dpp(0x7fffffff, pr, 0x130, 0xf, 0xf, false);   // wave_shl:1: lane c sees lane c + d
        vr_lo = __builtin_amdgcn_update_dpp(0, vr_lo, 0x130, 0xf, 0xf, false);
        vr_hi = __builtin_amdgcn_update_dpp(0, vr_hi, 0x130, 0xf, 0xf, false);
        const bool in_l = on && pos - pl <= md, in_r = on && pr - pos <= md;
        if (!__any(in_l || in_r)) break;
        if (d > 31) return false;
        if (in_l && __hiloint2double(vl_hi, vl_lo) > v) hl |= 1u << d;
        if (in_r && __hiloint2double(vr_hi, vr_lo) >= v) hr |= 1u << d;
        dmax = d;
    }
    int st = on ? 0 : 2;
    for (;;) {
        int sl = st, sr = st;
        bool any_kept = false, any_open = false;
        for (int d = 1; d <= dmax; ++d) {
            sl = __builtin_amdgcn_update_dpp(2, sl, 0x138, 0xf, 0xf, false);
            sr = __builtin_amdgcn_update_dpp(2, sr, 0x130, 0xf, 0xf, false);
            if ((hl >> d) & 1u) {
                any_kept |= sl == 1;
                any_open |= sl == 0;
            }
            if ((hr >> d) & 1u) {
                any_kept |= sr == 1;
                any_open |= sr == 0;
            }
        }
        if (st == 0) st = any_kept ? 2 : (any_open ? 0 : 1);
        if (!__any(st == 0)) break;
    }
    if (on) state[lane] = st;
    return true;
}

// The work list of the gaussian fits (round 6: sharded).  Until round 6 every frame reserved its places with two returning
// atomics on two counters of ONE cache line: 360 448 atomics per clip batch on one line of the memory-side atomic unit, ~11 ns each --
// 22 ns per frame against the 25 ns a frame of sacf_pfa_kernel takes altogether; without them (development ablation,
// profiles/r6/pfa_atomics_ablation.txt) the kernel ran 4.70 -> 3.94 ms at N = 2046 and 4.02 -> 2.9 ms at N = 1023.  Now ONE 64-bit
// atomic per frame (long count | other count << 32) on one of WL_SHARDS counters, each in a cache line of its own; a shard's items
// live in a region of their own (long fits from its front, the others from its back), and the fit kernel maps its running item
// number onto the shards through their prefix sums (peakfit_kernel).  Which fit is fetched when is scheduling: nothing computed
// depends on it.
constexpr int WL_SHARDS = 32, WL_SHARD_STRIDE = 16;   // counters: 16 x 8 B = 128 bytes apart

template <int T, bool WIDE = false>
__device__ __forceinline__ void peak_pick(const SacfArgs& a, long long f, double* yv, char* scratch, int tid) {
    typedef unsigned long long u64;
    constexpr int NW = T / 64;
    const int Mh = a.Mh, D = Mh - 1;  // D = len(dy)
    const int lane = tid & 63, wave = tid >> 6;
    u64* nzw = reinterpret_cast<u64*>(scratch);
    const int PW = peak_words(a.Mh);
    u64* candw = nzw + PW;
    u64* keptw = candw + PW;
    u64* longw = keptw + PW;
    int* cand = reinterpret_cast<int*>(longw + PW);  // [Mh/2 + 2]
    volatile int* state = cand + (Mh / 2 + 2);               // [Mh/2 + 2]: 0 undecided, 1 kept, 2 removed
    __shared__ double sh_red[2 * NW];
    __shared__ int sh_base[2];
    const u64 lt_mask = (1ull << lane) - 1ull;
    // ---- threshold, and the words of "dy[i] != 0" in the same pass over the row (round 6: one pass and one barrier fewer -- the
    //      barrier inside block_minmax publishes the words)
    const int E = (Mh + T - 1) / T, nwords = E * NW;
    double mx = -INFINITY, mn = INFINITY;
    for (int e = 0; e < E; ++e) {
        const int i = tid + e * T;
        const double v = yv[i < Mh ? i : 0], vn = yv[i < D ? i + 1 : 0];
        mx = i < Mh && v > mx ? v : mx;
        mn = i < Mh && v < mn ? v : mn;
        const u64 b = __ballot(i < D && vn != v);
        if (lane == 0) nzw[e * NW + wave] = b;
    }
    block_minmax<T>(sh_red, mx, mn);
    const double thres = a.peak_thresh * (mx - mn) + mn;
    if (D <= 0 || !__any((lane < nwords && nzw[lane < nwords ? lane : 0] != 0) ||
                         (WIDE && lane + 64 < nwords && nzw[lane + 64 < nwords ? lane + 64 : 0] != 0))) {
        if (tid == 0) a.peak_count[f] = 0;  // totally flat signal: no peaks (peakutils returns [])
        return;
    }
    // sign of the plateau-filled dy[i]: a zero run takes the value right after it (leading run), before it
    // (trailing run), or left / right of its median (interior run)
    auto dsign = [&](int i) -> int {
        double d = yv[i + 1] - yv[i];
        if (!(d != 0.0) || !(yv[i + 1] != yv[i])) {
            int w = i >> 6;
            u64 m = nzw[w] & ((1ull << (i & 63)) - 1ull);
            while (m == 0 && w > 0) m = nzw[--w];
            const int l = m ? w * 64 + 63 - __clzll((long long)m) : -1;  // nearest non-zero on the left
            w = i >> 6;
            m = (i & 63) == 63 ? 0ull : nzw[w] & (~0ull << ((i & 63) + 1));
            while (m == 0 && w < nwords - 1) m = nzw[++w];
            const int r = m ? w * 64 + __ffsll((long long)m) - 1 : D;    // ... on the right
            const int s0 = l + 1, e0 = r - 1;                            // the zero run containing i
            const int src = s0 == 0 ? r : (e0 == D - 1 ? l : (2 * i < s0 + e0 ? l : r));
            d = yv[src + 1] - yv[src];
        }
        return d > 0.0 ? 1 : (d < 0.0 ? -1 : 0);
    };
    // ---- candidates, ascending
    for (int e = 0; e < E; ++e) {
        const int i = tid + e * T;
        bool c = false;
        if (i >= 1 && i <= Mh - 2 && yv[i] > thres) c = dsign(i - 1) > 0 && dsign(i) < 0;
        const u64 b = __ballot(c);
        if (lane == 0) candw[e * NW + wave] = b;
    }
    __syncthreads();
    int ncand;
    {
        const int cnt = lane < nwords ? __popcll(candw[lane < nwords ? lane : 0]) : 0;
        const int cnt_hi = WIDE && lane + 64 < nwords ? __popcll(candw[lane + 64 < nwords ? lane + 64 : 0]) : 0;
        const WordScan<WIDE> scan(cnt, cnt_hi, lane);
        ncand = scan.total;
        for (int e = 0; e < E; ++e) {
            const int base = scan.word_base(e * NW + wave);
            const u64 b = candw[e * NW + wave];
            if ((b >> lane) & 1ull) {
                const int rank = base + __popcll(b & lt_mask);
                cand[rank] = tid + e * T;
                state[rank] = 0;
            }
        }
    }
    __syncthreads();
    // ---- minimum-distance suppression
    const int Q = (ncand + T - 1) / T, md = a.peak_min_dist;
    if (ncand > 1 && md > 1) {
        // a frame has a few dozen candidates: wave 0 runs the rounds alone, ordered by its own in-order LDS queue
        // instead of a workgroup barrier per round
        if (wave == 0 && ncand <= 64 && peak_rounds_in_lanes(cand, yv, state, ncand, md, lane)) {
            // (decided in registers: see peak_rounds_in_lanes)
        } else if (wave == 0) {
            const int Q0 = (ncand + 63) / 64;
            for (;;) {
                bool undecided = false;
                for (int q = 0; q < Q0; ++q) {
                    const int c = lane + q * 64;
                    if (c < ncand && state[c] == 0) {
                        const int pos = cand[c];
                        const double v = yv[pos];
                        bool any_kept = false, any_open = false;
                        for (int c2 = c - 1; c2 >= 0; --c2) {
                            const int p2 = cand[c2];
                            if (pos - p2 > md) break;
                            if (yv[p2] > v) {  // ties go to the larger index
                                const int st = state[c2];
                                any_kept |= st == 1;
                                any_open |= st == 0;
                            }
                        }
                        for (int c2 = c + 1; c2 < ncand; ++c2) {
                            const int p2 = cand[c2];
                            if (p2 - pos > md) break;
                            if (yv[p2] >= v) {
                                const int st = state[c2];
                                any_kept |= st == 1;
                                any_open |= st == 0;
                            }
                        }
                        if (any_kept) state[c] = 2;
                        else if (!any_open) state[c] = 1;
                        else undecided = true;
                    }
                }
                wave_lds_fence();
                if (!__any(undecided)) break;
            }
        }
        __syncthreads();
    } else {
        for (int c = tid; c < ncand; c += T) state[c] = 1;
        __syncthreads();
    }
    // ---- kept peaks -> peak_idx / work list.  Fits whose window maximum is not the peak sample itself (or
    // whose window is cut by the end of the lag range) are the ones that run away and burn MINPACK's whole
    // maxfev budget (measured: 93 % of the fits above 200 evaluations, 15 % of all fits).  They are queued at
    // the FRONT of the work list so that the persistent fit kernel starts them first and does not end on a
    // tail of stragglers; the others fill the list from the back.
    const int cwords = Q * NW;
    for (int q = 0; q < Q; ++q) {
        const int c = tid + q * T;
        const bool kept = c < ncand && state[c] == 1;
        bool lng = false;
        if (kept) {
            const int i = cand[c];
            const int lo = i - 10 > 0 ? i - 10 : 0, hi = i + 11 < Mh ? i + 11 : Mh;
            double mxw = yv[lo];
            for (int n = lo + 1; n < hi; ++n) mxw = yv[n] > mxw ? yv[n] : mxw;
            // ... and (round 6) the peaks with a window EDGE at 0.78 of the peak's height or more: 96 % of the runaway fits the
            // first rule misses (a slope that keeps rising out of the window), 18 % of the other peaks.  Only the ORDER of the
            // work list depends on this: every fit that may run away starts in the kernel's first trips.
            lng = mxw > yv[i] || hi - lo < 21 || fmax(yv[lo], yv[hi - 1]) >= 0.78 * yv[i];
        }
        const u64 bk = __ballot(kept), bl = __ballot(lng);
        if (lane == 0) {
            keptw[q * NW + wave] = bk;
            longw[q * NW + wave] = bl;
        }
    }
    __syncthreads();
    const int cnt2 = lane < cwords ? __popcll(keptw[lane < cwords ? lane : 0]) | (__popcll(longw[lane < cwords ? lane : 0]) << 16) : 0;
    const int cnt2_hi = WIDE && lane + 64 < cwords
                            ? __popcll(keptw[lane + 64 < cwords ? lane + 64 : 0]) | (__popcll(longw[lane + 64 < cwords ? lane + 64 : 0]) << 16)
                            : 0;
    const WordScan<WIDE> scan2(cnt2, cnt2_hi, lane);
    const int tot = scan2.total;
    int n_kept = tot & 0xffff;
    const int n_long = tot >> 16;
    int* out = a.peak_idx + f * (long long)a.maxp;
    if (n_kept > a.maxp) {  // cannot happen for maxp as esacf_run sizes it; serial and unsorted if it ever does
        if (tid == 0) {
            int p = 0;
            for (int c = 0; c < ncand && p < a.maxp; ++c)
                if (state[c] == 1) out[p++] = cand[c];
            a.peak_count[f] = p;
            const int sh = (int)(f % WL_SHARDS);
            const unsigned long long old = atomicAdd(a.shard_cnt + sh * WL_SHARD_STRIDE, (unsigned long long)p << 32);
            const int back = (sh + 1) * a.shard_cap - 1 - (int)(old >> 32);
            for (int j = 0; j < p; ++j) a.worklist[back - j] = (int)(f << 12) | j;
        }
        return;
    }
    if (tid == 0) {
        a.peak_count[f] = n_kept;
        const int sh = (int)(f % WL_SHARDS);
        if (a.ablate & 32) {   // (development ablation: what the returning atomic on the work list's counters costs -- results are garbage)
            sh_base[0] = sh * a.shard_cap;
            sh_base[1] = (sh + 1) * a.shard_cap - 1;
        } else if (n_kept) {
            const unsigned long long old = atomicAdd(a.shard_cnt + sh * WL_SHARD_STRIDE,
                                                     (unsigned long long)(unsigned)n_long | ((unsigned long long)(unsigned)(n_kept - n_long) << 32));
            sh_base[0] = sh * a.shard_cap + (int)(unsigned)old;
            sh_base[1] = (sh + 1) * a.shard_cap - 1 - (int)(old >> 32);
        }
    }
    __syncthreads();
    for (int q = 0; q < Q; ++q) {
        const int base = scan2.word_base(q * NW + wave);
        const u64 bk = keptw[q * NW + wave], bl = longw[q * NW + wave];
        if ((bk >> lane) & 1ull) {
            const int p = (base & 0xffff) + __popcll(bk & lt_mask);
            const int pl = (base >> 16) + __popcll(bl & lt_mask);
            out[p] = cand[tid + q * T];
            const int item = (int)(f << 12) | p;
            if ((bl >> lane) & 1ull) a.worklist[sh_base[0] + pl] = item;
            else a.worklist[sh_base[1] - (p - pl)] = item;
        }
    }
}

// |X|^0.67 of one band (esacf.py:95-103: np.abs(X) ** 0.67) as exp(0.335 * log(|X|^2)): two short libm
// polynomials instead of hypot + the general pow (a third of the kernel's time before).  The spectra of
// audio-range input are far from the overflow/underflow guards of hypot; the result is within ~15 ulp of
// numpy's (|0.335 ln q| <= 30 amplifies the rounding of log), two orders of magnitude below the rounding
// noise the FFTs put on the SACF.  |X| = 0 gives exp(-inf) = 0 like 0 ** 0.67.
// |X|^0.67 of one band (esacf.py:95-101, k fixed at 0.67) from its real and imaginary part; tab: mpx_pow067.hpp tables in LDS
__device__ __forceinline__ double mag067(double re, double im, const double* tab) { return p067::pow067(re * re + im * im, tab); }
// the library form (until round 3 everywhere; still sacf_pfa_kernel<1>, see there)
__device__ __forceinline__ double mag067_libm(double re, double im) { return exp(0.335 * log(re * re + im * im)); }
// every SACF kernel keeps the 193-entry table in (static) LDS: filled at the start, first used behind a workgroup barrier
#define MPX_POW067_LDS(a, tid, nthreads)                                   \
    __shared__ double pow_tab[p067::TAB_DOUBLES];                           \
    for (int i_ = (tid); i_ < p067::TAB_DOUBLES; i_ += (nthreads)) pow_tab[i_] = (a).pow_tab[i_]

// The truncation regime of esacf.py:108-129 (every rate's "stretched" copy is a prefix of the row: < 1024 lags, or the no-op mode) in
// ONE round.  Round r = 2 .. n_peaks_elim of the reference clips, subtracts the first round(Mh / r) lags from themselves (v - v:
// NaN and inf become NaN like x - x does), clips.  The prefixes are NESTED -- round-half-even(Mh / r) does not grow with r -- and
// clipping is idempotent: after round 2, the longest prefix, a lag is 0 (or NaN) inside it and max(v, 0) outside, and rounds
// 3 .. n_peaks_elim change nothing (0 - 0 = 0, NaN stays).  Until round 6 every lag ran all the rounds, a double-precision DIVISION
// each (Mh / r): 20 divisions per thread and frame of 2046 samples, 18 % of sacf_pfa_kernel<2>'s VALU instructions.
// cut2 = enhance_cut2(...) once per thread: round(Mh / 2) (0.5 Mh is exact), 0 in the no-op mode.
__device__ __forceinline__ int enhance_cut2(int Mh, int enhance_mode) {
    return enhance_mode == MPX_ENHANCE_LIBROSA010 ? (int)nearbyint(0.5 * (double)Mh) : 0;
}
__device__ __forceinline__ double enhance_truncate(double v, int n, int n_peaks_elim, int cut2) {
    if (n_peaks_elim >= 2) {
        v = v < 0.0 ? 0.0 : v;          // clip
        if (n < cut2) v = v - v;        // minus the "stretched" copy (== itself for a <=2-frame STFT)
        v = v < 0.0 ? 0.0 : v;          // clip
    }
    return v;
}

template <int L, bool BLUE>
__global__ __launch_bounds__(L / 8, 4) void sacf_kernel(SacfArgs a) {
    constexpr int T = L / 8;
    using PL = DifPlan<L>;
    constexpr int RL = PL::radix(PL::n - 1);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = a.N, Mh = a.Mh;
    cx<double>* buf = reinterpret_cast<cx<double>*>(smem);                      // L complex, slot sigma<L>(position)
    double* yv = reinterpret_cast<double*>(smem + sizeof(cx<double>) * L);      // Mh + 2 doubles
    const int tid = threadIdx.x;
    MPX_POW067_LDS(a, tid, T);
    const DifTwiddles<L, double> twd = dif_load_twiddles<L, double>(a.tw, tid);
    cx<double> regs[8];
    cx<double> chv[4];  // this thread's chirp values (Bluestein)
#pragma unroll
    for (int r = 0; r < 4; ++r) chv[r] = BLUE ? a.chirp[(tid + r * (L / 8)) < N ? tid + r * (L / 8) : 0] : cx<double>{1.0, 0.0};

    // ---- SACF: DFT_N(x_lo + i x_hi) -> S -> DFT_N(S) -> first Mh lags / N.
    // With a.pair (experimental, off by default: see esacf_run) a workgroup takes TWO frames: S is real and even, so DFT_N(S) is real, and one complex transform of
    // S_a + i S_b returns the lags of frame a in its real part and those of frame b in its imaginary part
    // (the cross-talk is the rounding-level imaginary part of a real-even DFT): 3 DFTs per 2 frames instead of 4.
    const long long fa = a.pair ? 2 * (long long)blockIdx.x : (long long)blockIdx.x;
    const bool have_b = a.pair && fa + 1 < a.num_frames;
    double* sa_half = yv;                       // S_a[0 .. N/2] waits here while frame b's spectrum is computed
    double* sd = reinterpret_cast<double*>(buf);  // S of the frame in flight, all N bins (aliases buf)
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        if (h == 1 && !have_b) break;
        // (an opaque copy of the thread id per trip: everything below depends only on the thread, and a compiler
        //  that hoists all of it out of this two-trip loop runs out of registers and spills)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        int kk[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) kk[e] = BLUE ? tid + e * T : dif_freq<L>(dif_last_pos<L>(tid, e / RL, e % RL));
        const cx<double>* xin = a.xb + band_index(fa + h, 0, N);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int n = tid + r * T;
            regs[r] = n < N ? xin[(size_t)(n >> 4) * (64 * BS_TILE) + (n & 15)] : cx<double>{0.0, 0.0};
        }
        if (!(a.ablate & 8)) dft_regs<L, BLUE>(buf, twd, a, regs, tid, chv);
        // every thread needs the mirror bin X[N-k] of each of its bins: exchange through LDS
#pragma unroll
        for (int e = 0; e < 8; ++e) buf[sigma<L>(BLUE ? kk[e] : dif_last_pos<L>(tid, e / RL, e % RL))] = regs[e];
        __syncthreads();
        // S[k] = |X_lo[k]|^0.67 + |X_hi[k]|^0.67 is even in k (the bands are real): the owner of k <= N/2 computes
        // it for N-k as well (bit-identical: the mirror pair only swaps the operands of commutative adds and flips
        // signs that the squares drop), so the two exp/log per bin run on half the bins.
        double sv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = kk[e];
            sv[e] = 0.0;
            if (2 * k <= N) {
                const int km = k == 0 ? 0 : N - k;
                const cx<double> A = regs[e];
                const cx<double> B = cconj(buf[sigma<L>(BLUE ? km : dif_pos<L>(km))]);
                // X_lo = (A + B)/2 ; X_hi = (A - B)/(2i)
                const double lr = 0.5 * (A.x + B.x), li = 0.5 * (A.y + B.y);
                const double hr = 0.5 * (A.y - B.y), hm = -0.5 * (A.x - B.x);
                sv[e] = (a.ablate & 1) ? lr + hr : mag067(lr, li, pow_tab) + mag067(hr, hm, pow_tab);  // k fixed at 0.67
            }
            __builtin_amdgcn_sched_barrier(0);   // one bin (two powers) at a time: interleaved, the eight of them spill
        }
        __syncthreads();  // mirror reads done: buf is free
        if (h == 0 && have_b) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (2 * kk[e] <= N) sa_half[kk[e]] = sv[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = kk[e];
                if (2 * k <= N) {
                    sd[k] = sv[e];
                    if (k != 0) sd[N - k] = sv[e];
                }
            }
        }
        // (the next writers of buf are this loop's second DFT or the transform below, both behind a barrier)
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = BLUE ? tid + e * T : dif_freq<L>(dif_last_pos<L>(tid, e / RL, e % RL));
        if (k >= N) regs[e] = {0.0, 0.0};
        else if (have_b) regs[e] = {sa_half[2 * k <= N ? k : N - k], sd[k]};
        else regs[e] = {sd[k], 0.0};
    }
    __syncthreads();  // S reads done before the transform writes buf
    if (a.ablate & 4) {
    } else if (BLUE)
        dft_regs<L, true>(buf, twd, a, regs, tid, chv);
    else
        idit_fft_from_last<L, double>(buf, twd, regs, tid);  // S is real and even: its inverse DFT x N is its DFT
    __syncthreads();  // buf is dead from here on: frame b's lags and the peak-picking scratch alias it
    double* yvb = reinterpret_cast<double*>(smem + sizeof(cx<double>) * (L / 2));
    const double inv_n = 1.0 / (double)N;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int n = tid + r * T;
        if (n < Mh) {
            const double va = regs[r].x * inv_n, vb = regs[r].y * inv_n;
            yv[n] = va;
            yvb[n] = vb;
            if (a.sacf_out) {
                a.sacf_out[fa * (long long)Mh + n] = va;
                if (have_b) a.sacf_out[(fa + 1) * (long long)Mh + n] = vb;
            }
        }
    }
    __syncthreads();
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        if (h == 1 && !have_b) break;
        const long long f = fa + h;
        double* yh = h ? yvb : yv;
        double* yrow = a.y_out + f * (long long)Mh;
        if (a.defer_enhance) {  // phase-vocoder regime: pv_enhance_kernel + peakpick_kernel take over from the raw SACF
            for (int n = tid; n < Mh; n += T) yrow[n] = yh[n];
            continue;
        }
        // ---- enhancement (esacf.py:108-129)
        // (every lag is touched by one thread only, the same one in every pass: registers, one store, one barrier)
        const int cut2 = enhance_cut2(Mh, a.enhance_mode);
        for (int n = tid; n < Mh; n += T) {
            const double v = enhance_truncate(yh[n], n, a.n_peaks_elim, cut2);
            yh[n] = v;
            yrow[n] = v;
        }
        __syncthreads();
        if (a.ablate & (2 | 16)) continue;  // 16: peak picking runs as its own one-wave-per-frame kernel
        peak_pick<T>(a, f, yh, smem, tid);
        __syncthreads();  // frame a's scratch is dead before frame b's
    }
}

// Non-power-of-two frames of 2049 ... 2730 samples (the reference's default 46.4 ms frame at 48 kHz: 2227): a full chirp-z
// needs 2N - 1 > 4096 points, i.e. the 8192-point Stockham path at one workgroup per CU (sacf_big_kernel, ~4x the time per
// frame).  But the SACF only ever looks at HALF of every transform: the bands are real, so |X[k]| is needed for k <= N/2
// only, and the lags are the real part of the forward DFT of the even sequence S, r[n] = Re sum_{k <= N/2} w_k S[k] W_N^(kn) / N
// (w = 1 at k = 0 and N/2, else 2), for n < N/2.  A chirp-z evaluated at K = N/2 + 1 bins is a circular convolution of
// N + K - 1 ~ 1.5 N points: 4096 do.  So a frame is THREE real chirp-z transforms -- x_lo, x_hi, w S -- on the in-place
// engine, all three with the same chirp and the same filter spectrum; the |.|^0.67 needs no final chirp multiplication (it
// drops the phase), and bin k of the first two and sample k of the third belong to the same thread: no exchange through LDS.
template <int L>
__global__ __launch_bounds__(L / 8, 2) void sacf_rz_kernel(SacfArgs a) {
    constexpr int T = L / 8;
    static_assert(L == 4096, "N <= 2730 < 6 T inputs and K <= 1366 < 3 T outputs per thread are written for L = 4096");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = a.N, Mh = a.Mh, K = N / 2 + 1;
    cx<double>* buf = reinterpret_cast<cx<double>*>(smem);                      // L complex, slot sigma<L>(position)
    double* yv = reinterpret_cast<double*>(smem + sizeof(cx<double>) * L);      // Mh + 2 doubles
    const int tid = threadIdx.x;
    MPX_POW067_LDS(a, tid, T);
    const DifTwiddles<L, double> twd = dif_load_twiddles<L, double>(a.tw, tid);
    const long long f = blockIdx.x;
    cx<double> regs[8];
    double hi[6];
    const cx<double>* xin = a.xb + band_index(f, 0, N);
    // length-L circular convolution of regs (natural order in, natural order out) with the chirp filter
    auto convolve = [&]() {
        dif_fft_keep_last<L, double>(buf, twd, regs, tid);
#pragma unroll
        for (int e = 0; e < 8; ++e) regs[e] = cmul(regs[e], a.bhat[e * T + tid]);
        idit_fft_from_last<L, double>(buf, twd, regs, tid);
    };
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int n = tid + r * T;
        regs[r] = {0.0, 0.0};
        if (r < 6) {
            const bool in = n < N;
            const cx<double> v = xin[in ? (size_t)(n >> 4) * (64 * BS_TILE) + (n & 15) : 0];
            const cx<double> ch = a.chirp[in ? n : 0];
            hi[r] = in ? v.y : 0.0;
            const double lo = in ? v.x : 0.0;
            regs[r] = {lo * ch.x, -lo * ch.y};   // x_lo[n] conj(chirp[n])
        }
    }
    convolve();
    double sv[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) sv[r] = tid + r * T < K ? mag067(regs[r].x, regs[r].y, pow_tab) : 0.0;
    __syncthreads();   // every wave has left the transform before the next one writes buf
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int n = tid + r * T;
        regs[r] = {0.0, 0.0};
        if (r < 6) {
            const cx<double> ch = a.chirp[n < N ? n : 0];
            regs[r] = {hi[r] * ch.x, -hi[r] * ch.y};   // hi[r] is zero beyond the frame
        }
    }
    convolve();
#pragma unroll
    for (int r = 0; r < 3; ++r)
        if (tid + r * T < K) sv[r] += mag067(regs[r].x, regs[r].y, pow_tab);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int k = tid + r * T;
        regs[r] = {0.0, 0.0};
        if (r < 3 && k < K) {
            const double s = (k == 0 || 2 * k == N) ? sv[r] : 2.0 * sv[r];
            const cx<double> ch = a.chirp[k];
            regs[r] = {s * ch.x, -s * ch.y};
        }
    }
    convolve();
    const double inv_n = 1.0 / (double)N;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int n = tid + r * T;
        if (n < Mh) {
            const cx<double> ch = a.chirp[n];
            const double v = (regs[r].x * ch.x + regs[r].y * ch.y) * inv_n;   // Re(y[n] conj(chirp[n])) / N
            yv[n] = v;
            if (a.sacf_out) a.sacf_out[f * (long long)Mh + n] = v;
        }
    }
    __syncthreads();  // buf is dead from here on; the peak-picking scratch aliases it
    double* yrow = a.y_out + f * (long long)Mh;
    if (a.defer_enhance) {
        for (int n = tid; n < Mh; n += T) yrow[n] = yv[n];
        return;
    }
    const int cut2 = enhance_cut2(Mh, a.enhance_mode);
    for (int n = tid; n < Mh; n += T) {   // enhancement without the vocoder (no-op mode, or fewer than two rates): see sacf_kernel
        const double v = enhance_truncate(yv[n], n, a.n_peaks_elim, cut2);
        yv[n] = v;
        yrow[n] = v;
    }
    __syncthreads();
    peak_pick<T>(a, f, yv, smem, tid);
}

// ------------------------------------------------------------------ kernel 2, prime-factor engine
// The reference's own frame lengths are products of small coprimes: 1023 = 3 * 11 * 31 (46.4 ms at 22.05 kHz) and
// 2046 = 2 * 3 * 11 * 31 (44.1 kHz).  Good-Thomas: with the input index written as n = sum_i n_i N/N_i (mod N) and the
// output index by its residues k_i = k mod N_i, W_N^(nk) = prod_i W_(N_i)^(n_i k_i): the N-point DFT IS a multi-dimensional
// DFT over the array [A0][3][11][31] -- no twiddle factors, no zero padding (the chirp-z path runs four 4096-point
// transforms per 2046-sample frame).  And the same array DFT maps residue-indexed input to sum-indexed output, so the
// second transform of the SACF (DFT of S, which sits at residue positions after the first) runs in place on the same
// layout and leaves lag n at the position sample n was loaded to: one position table serves both ends.
//   axis 31, axis 11: one line per 16 / 6 lanes; lane k holds cos/sin(2 pi n k / P) for its k in registers and computes the
//     output pair X[k], X[P-k] from the line's sums s[n] = a[n] + a[P-n] and differences d[n] = a[n] - a[P-n]:
//     X[k], X[P-k] = a0 + sum_n s[n] cos(nk) -+ i sum_n d[n] sin(nk): (P-1)/2 * 4 fused multiply-adds per pair instead
//     of 4 (P-1) for a plain matrix row.  The lanes of a line sit in one wave: the passes are ordered by the wave's own
//     LDS queue, workgroup barriers only between axes.
//   axes A0 x 3: one thread per line, a 6- (or 3-) point DFT in registers.
constexpr int PFA_T = 256;
// LDS map of sacf_pfa_kernel: [buf: N complex][cs31: 256 complex][cs11: 30 complex]; once the second transform has been
// read out, buf is dead and holds [peak-picking scratch][yv: Mh + 2 doubles].  37.3 KB at N = 2046: four workgroups per CU
// (with the lags and the position table in LDS as well it was 49.6 KB and three).
#define PFA_TAB_OFF(N, Mh) (16 * (size_t)(N))
#define PFA_LDS_BYTES(N, Mh) (PFA_TAB_OFF(N, Mh) + 16 * (256 + 30))
#define PFA_YV_OFF(Mh) ((peak_scratch_bytes(Mh) + 15) & ~(size_t)15)

// Axis 31: a line per 16-lane DPP row, lane k holds s[k] = a[k] + a[31-k] and d[k] = a[k] - a[31-k] (lane 0: a[0], 0)
// and needs all sixteen of each.  gfx950 lets a double-precision VOP2 take its first operand through DPP with
// row_newbcast:n -- lane n of the row, broadcast to the row -- so term n of every lane's sums is ONE instruction,
// v_fmac_f64_dpp acc, s (lane n), coefficient[n], and the coefficients are simply indexed by n.  (The first version read
// s and d back from LDS, 31 broadcast reads per lane: bound by the LDS pipe.  The second rotated them past with
// row_ror: two 32-bit v_mov_dpp per double and term, 8 moves + 4 multiply-adds where this has 4 multiply-adds.)
// Every lane reads and writes only its own two elements: no fences inside the pass.
struct Pfa31Coef {
    double c[16], s[16];   // (cos, sin)(2 pi n k / 31), n = 0..15, for this lane's k
};

// acc += (value of `v` in lane N of the caller's 16-lane row) * coef
template <int N>
__device__ __forceinline__ void fmac_row_bcast(double& acc, double v, double coef) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(coef), "n"(N));
}

template <int N>
__device__ __forceinline__ void pfa31_term(const cx<double>& sv, const cx<double>& dv, const Pfa31Coef& w, cx<double>& pp, cx<double>& qq) {
    fmac_row_bcast<N>(pp.x, sv.x, w.c[N]);
    fmac_row_bcast<N>(pp.y, sv.y, w.c[N]);
    fmac_row_bcast<N>(qq.x, dv.x, w.s[N]);
    fmac_row_bcast<N>(qq.y, dv.y, w.s[N]);
}

// Real input (the second transform of the SACF acts on S): the sums and differences are real, so are P and Q, and
// X[31-k] = conj X[k] -- half the multiply-adds, and only X[k], k <= 15, is stored: the later axes then run on the 16 of
// 31 residues that are kept (the caller reads lag n from its mirror position when it must).
template <int N>
__device__ __forceinline__ void pfa31_term_real(double sv, double dv, const Pfa31Coef& w, double& pp, double& qq) {
    fmac_row_bcast<N>(pp, sv, w.c[N]);
    fmac_row_bcast<N>(qq, dv, w.s[N]);
}

// cs31: LDS copy of the [16][16] table (cos, sin)(2 pi n k / 31) -- symmetric in n and k, read as row n, column k: the
// sixteen lanes of a row read sixteen neighbouring entries (as row k the reads are 256 B apart: a sixteen-way bank
// conflict on each).  A lane's sixteen entries are fetched at the start of a pass and live in registers only while it runs.
template <bool REAL>
__device__ __forceinline__ void pfa_pass31(cx<double>* buf, int nlines, const cx<double>* cs31, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int sub = lane >> 4, k = lane & 15;
    Pfa31Coef w;
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        const cx<double> v = cs31[n * 16 + k];
        w.c[n] = v.x;
        w.s[n] = v.y;
    }
    for (int l0 = 0; l0 < nlines; l0 += 16) {
        if (l0 + wave * 4 >= nlines) break;      // the last round of 66 lines has work for one wave only (wave-uniform)
        const int l = l0 + wave * 4 + sub;
        const bool active = l < nlines;           // uniform per 16-lane row
        cx<double>* e = buf + (active ? l : 0) * 31;
        const int ia = k, ib = k ? 31 - k : 0;
        if (REAL) {
            const double ur = e[ia].x, vr = e[ib].x;
            double sv = k ? ur + vr : ur, dv = k ? ur - vr : 0.0;
            // even and odd terms in separate sums: four dependent chains instead of two (the multiply-adds of one chain
            // are a result latency apart, and in the last round of a pass a single wave is at work)
            double pp = 0.0, qq = 0.0, pp1 = 0.0, qq1 = 0.0;
            // (a DPP operand must have been written two instructions earlier: the hazard is invisible to the compiler
            //  inside inline assembly)
            asm volatile("s_nop 1" : "+v"(sv), "+v"(dv));
            pfa31_term_real<0>(sv, dv, w, pp, qq);
            pfa31_term_real<1>(sv, dv, w, pp1, qq1);
            pfa31_term_real<2>(sv, dv, w, pp, qq);
            pfa31_term_real<3>(sv, dv, w, pp1, qq1);
            pfa31_term_real<4>(sv, dv, w, pp, qq);
            pfa31_term_real<5>(sv, dv, w, pp1, qq1);
            pfa31_term_real<6>(sv, dv, w, pp, qq);
            pfa31_term_real<7>(sv, dv, w, pp1, qq1);
            pfa31_term_real<8>(sv, dv, w, pp, qq);
            pfa31_term_real<9>(sv, dv, w, pp1, qq1);
            pfa31_term_real<10>(sv, dv, w, pp, qq);
            pfa31_term_real<11>(sv, dv, w, pp1, qq1);
            pfa31_term_real<12>(sv, dv, w, pp, qq);
            pfa31_term_real<13>(sv, dv, w, pp1, qq1);
            pfa31_term_real<14>(sv, dv, w, pp, qq);
            pfa31_term_real<15>(sv, dv, w, pp1, qq1);
            pp += pp1;
            qq += qq1;
            if (active) e[ia] = {pp, -qq};   // X[k] = P - iQ
            continue;
        }
        const cx<double> u = e[ia], v = e[ib];
        cx<double> sv = k ? cx<double>{u.x + v.x, u.y + v.y} : u;
        cx<double> dv = k ? cx<double>{u.x - v.x, u.y - v.y} : cx<double>{0.0, 0.0};
        cx<double> pp = {0.0, 0.0}, qq = {0.0, 0.0}, pp1 = {0.0, 0.0}, qq1 = {0.0, 0.0};
        asm volatile("s_nop 1" : "+v"(sv.x), "+v"(sv.y), "+v"(dv.x), "+v"(dv.y));
        pfa31_term<0>(sv, dv, w, pp, qq);
        pfa31_term<1>(sv, dv, w, pp1, qq1);
        pfa31_term<2>(sv, dv, w, pp, qq);
        pfa31_term<3>(sv, dv, w, pp1, qq1);
        pfa31_term<4>(sv, dv, w, pp, qq);
        pfa31_term<5>(sv, dv, w, pp1, qq1);
        pfa31_term<6>(sv, dv, w, pp, qq);
        pfa31_term<7>(sv, dv, w, pp1, qq1);
        pfa31_term<8>(sv, dv, w, pp, qq);
        pfa31_term<9>(sv, dv, w, pp1, qq1);
        pfa31_term<10>(sv, dv, w, pp, qq);
        pfa31_term<11>(sv, dv, w, pp1, qq1);
        pfa31_term<12>(sv, dv, w, pp, qq);
        pfa31_term<13>(sv, dv, w, pp1, qq1);
        pfa31_term<14>(sv, dv, w, pp, qq);
        pfa31_term<15>(sv, dv, w, pp1, qq1);
        pp = {pp.x + pp1.x, pp.y + pp1.y};
        qq = {qq.x + qq1.x, qq.y + qq1.y};
        if (active) {
            // X[k] = P - iQ, X[31-k] = P + iQ   (a[0] is the n = 0 term of P)
            e[ia] = {pp.x + qq.y, pp.y - qq.x};
            if (k) e[ib] = {pp.x - qq.y, pp.y + qq.x};
        }
    }
}

template <int P, int KL>   // P = 11, KL = (P + 1) / 2 lanes per line; csP: LDS table [KL][H] (cos, sin)(2 pi n k / P), n = 1..H
__device__ __forceinline__ void pfa_prime_pass(cx<double>* buf, int nlines, int inner, int outer_stride, int stride,
                                               const cx<double>* csP, int tid) {
    constexpr int H = (P - 1) / 2, LW = 64 / KL, LPR = LW * (PFA_T / 64);
    const int lane = tid & 63, wave = tid >> 6;
    const int sub = lane / KL, k = lane - sub * KL;
    double cosv[H], sinv[H];
#pragma unroll
    for (int n = 0; n < H; ++n) {
        const cx<double> v = csP[k * H + n];
        cosv[n] = v.x;
        sinv[n] = v.y;
    }
    for (int l0 = 0; l0 < nlines; l0 += LPR) {
        if (l0 + wave * LW >= nlines) break;     // wave-uniform
        const int l = l0 + wave * LW + sub;
        const bool active = sub < LW && l < nlines;
        // line l -> first element: lines are numbered (outer, inner) with `inner` consecutive positions
        const int base = active ? (l / inner) * outer_stride + (l % inner) : 0;
        cx<double>* e = buf + base;
        if (active && k >= 1) {
            const cx<double> u = e[k * stride], v = e[(P - k) * stride];
            e[k * stride] = {u.x + v.x, u.y + v.y};
            e[(P - k) * stride] = {u.x - v.x, u.y - v.y};
        }
        wave_lds_fence();
        cx<double> a0 = {0.0, 0.0}, pp = {0.0, 0.0}, qq = {0.0, 0.0};
        if (active) {
            a0 = e[0];
#pragma unroll
            for (int n = 1; n <= H; ++n) {
                const cx<double> sv = e[n * stride], dv = e[(P - n) * stride];
                pp.x = fma(sv.x, cosv[n - 1], pp.x);
                pp.y = fma(sv.y, cosv[n - 1], pp.y);
                qq.x = fma(dv.x, sinv[n - 1], qq.x);
                qq.y = fma(dv.y, sinv[n - 1], qq.y);
            }
        }
        wave_lds_fence();
        if (active) {
            // X[k] = a0 + P - iQ, X[P-k] = a0 + P + iQ
            const double br = a0.x + pp.x, bi = a0.y + pp.y;
            e[k * stride] = {br + qq.y, bi - qq.x};
            if (k >= 1) e[(P - k) * stride] = {br - qq.y, bi + qq.x};
        }
        wave_lds_fence();
    }
}

// y = DFT_3(x), forward
__device__ __forceinline__ void dft3(cx<double>& x0, cx<double>& x1, cx<double>& x2) {
    const double H3 = 0.8660254037844386467637;  // sqrt(3)/2
    const cx<double> t = {x1.x + x2.x, x1.y + x2.y}, u = {x1.x - x2.x, x1.y - x2.y};
    const cx<double> m = {fma(-0.5, t.x, x0.x), fma(-0.5, t.y, x0.y)};
    x0 = {x0.x + t.x, x0.y + t.y};
    // -i (sqrt3/2) u = ((sqrt3/2) u.y, -(sqrt3/2) u.x)
    x1 = {fma(H3, u.y, m.x), fma(-H3, u.x, m.y)};
    x2 = {fma(-H3, u.y, m.x), fma(H3, u.x, m.y)};
}

template <int A0, int K3>   // K3: residues of axis 31 that are live (31, or 16 behind a real-input pass)
__device__ __forceinline__ void pfa_small_pass(cx<double>* buf, int tid) {
    constexpr int Q = 11 * 31;   // lines; element j = n0 * 3 + n1 of line q at j * Q + q
    for (int idx = tid; idx < 11 * K3; idx += PFA_T) {
        const int q = K3 == 31 ? idx : (idx / K3) * 31 + idx % K3;
        cx<double> v[3 * A0];
#pragma unroll
        for (int j = 0; j < 3 * A0; ++j) v[j] = buf[j * Q + q];
#pragma unroll
        for (int n0 = 0; n0 < A0; ++n0) dft3(v[3 * n0], v[3 * n0 + 1], v[3 * n0 + 2]);
        if (A0 == 2) {
#pragma unroll
            for (int k1 = 0; k1 < 3; ++k1) {
                const cx<double> a = v[k1], b = v[3 + k1];
                v[k1] = {a.x + b.x, a.y + b.y};
                v[3 + k1] = {a.x - b.x, a.y - b.y};
            }
        }
#pragma unroll
        for (int j = 0; j < 3 * A0; ++j) buf[j * Q + q] = v[j];
    }
}

template <int A0, bool REAL>
__device__ __forceinline__ void pfa_dft(cx<double>* buf, const cx<double>* cs31, const cx<double>* cs11, int tid) {
    constexpr int K3 = REAL ? 16 : 31;
    pfa_pass31<REAL>(buf, A0 * 3 * 11, cs31, tid);                               // line l: elements l*31 + n3
    __syncthreads();
    pfa_prime_pass<11, 6>(buf, A0 * 3 * K3, K3, 11 * 31, 31, cs11, tid);           // line (m, n3): m*341 + n3 + 31 n2
    __syncthreads();
    pfa_small_pass<A0, K3>(buf, tid);
    __syncthreads();
}

// One workgroup per frame; the cosine/sine tables and the position table go to LDS first.
template <int A0>
__global__ __launch_bounds__(PFA_T, 4) void sacf_pfa_kernel(SacfArgs a) {
    constexpr int T = PFA_T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // The frame length is a compile-time constant of the instantiation (the host launches <A0> for N = A0 * 1023 only): the
    // gather, pair and lag loops below unroll, so that their table loads, sample loads and LDS reads are all in flight
    // together instead of one dependent global round trip per iteration.
    constexpr int N = A0 * 3 * 11 * 31, Mh = (N - 1) / 2;
    cx<double>* buf = reinterpret_cast<cx<double>*>(smem);                      // [A0][3][11][31]
    double* yv = reinterpret_cast<double*>(smem + PFA_YV_OFF(Mh));              // Mh + 2 doubles, inside the dead buf
    const unsigned short* __restrict__ pos = a.pfa_pos;                         // [N] position of index n (L2-resident)
    cx<double>* cs31 = reinterpret_cast<cx<double>*>(smem + PFA_TAB_OFF(N, Mh));   // [16][16]
    cx<double>* cs11 = cs31 + 256;                                                    // [6][5]
    __shared__ double pow_tab[p067::TAB_DOUBLES];
    // all three table loads are issued before the first of them is stored, and (round 6) the frame's own loads with them: one
    // L2 / HBM round trip and ONE barrier in front of the first transform, not two of each
    const cx<double> v31 = a.pfa_cs31[threadIdx.x];
    const cx<double> v11 = a.pfa_cs11[threadIdx.x < 30 ? threadIdx.x : 0];
    const double vp = A0 == 1 ? 0.0 : a.pow_tab[threadIdx.x < p067::TAB_DOUBLES ? threadIdx.x : 0];
    const double inv_n = 1.0 / (double)N;
    {
        // (one workgroup per frame: as persistent workgroups looping over frames the compiler hoisted the loop-invariant
        //  addresses of everything below -- peak picking alone has dozens -- and spilled 350 B per lane; the tables a
        //  workgroup loads above are 8.6 KB from L2 against 32 KB of frame data)
        const long long f = blockIdx.x;
        const int tid = threadIdx.x;
        const cx<double>* xin = a.xb + band_index(f, 0, N);
        constexpr int NPAIRS = N / 2 + 1, PPER = (NPAIRS + T - 1) / T;   // mirror pairs k <= N - k: k = 0 .. N/2 (host: pfa_npairs)
        unsigned pairq[PPER];    // this thread's pairs (position of k | position of N - k << 16): loaded with the gather, used behind the first transform
        int plag[4], pmir[4];   // positions of the lags n = tid + r T, r < 4 (n < Mh), and of their mirrors N - n: read with the
                                // gather's own table loads (until round 6: eight dependent L2 loads per thread in front of the lags)
        {
            constexpr int PER = (N + T - 1) / T;
            cx<double> xv[PER];
            int pv[PER];
#pragma unroll
            for (int r = 0; r < PER; ++r) {
                const int n = tid + r * T, nc = n < N ? n : 0;
                pv[r] = pos[nc];
                xv[r] = xin[(size_t)(nc >> 4) * (64 * BS_TILE) + (nc & 15)];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = tid + r * T;
                pmir[r] = pos[n >= 1 && n < Mh ? N - n : 0];
            }
#pragma unroll
            for (int r = 0; r < PPER; ++r) {
                const int i = tid + r * T;
                pairq[r] = reinterpret_cast<const unsigned*>(a.pfa_pairs)[i < NPAIRS ? i : 0];
            }
            cs31[tid] = v31;
            if (tid < 30) cs11[tid] = v11;
            if (A0 != 1 && tid < p067::TAB_DOUBLES) pow_tab[tid] = vp;
#pragma unroll
            for (int r = 0; r < PER; ++r)
                if (tid + r * T < N) buf[pv[r]] = xv[r];
#pragma unroll
            for (int r = 0; r < 4; ++r) plag[r] = r < PER ? pv[r] : 0;
        }
        __syncthreads();
        if (!(a.ablate & 8)) pfa_dft<A0, false>(buf, cs31, cs11, tid);
        // S[k] = |X_lo[k]|^0.67 + |X_hi[k]|^0.67 from Z[k] and its mirror bin Z[N-k]: one thread per pair, both positions
        // get the (real, even) value; nobody else touches the pair
        {
            constexpr int PER = PPER;
            int pp[PER], qq[PER];
#pragma unroll
            for (int r = 0; r < PER; ++r) {
                pp[r] = (int)(pairq[r] & 0xffffu);
                qq[r] = (int)(pairq[r] >> 16);
            }
#pragma unroll
            for (int r = 0; r < PER; ++r) {
                const cx<double> A = buf[pp[r]];
                const cx<double> B = cconj(buf[qq[r]]);
                const double lr = 0.5 * (A.x + B.x), li = 0.5 * (A.y + B.y);
                const double hr = 0.5 * (A.y - B.y), hm = -0.5 * (A.x - B.x);
                // N = 1023 (two pairs per thread: four powers in flight) is FASTER on the library's log / exp, which never
                // wait for a table read (3.9 against 4.3 ms per 176 k frames); N = 2046 (eight in flight) on the tables
                // (6.0 -> 5.6 ms).  Both are within their error of the true power (4e-15 / 3e-16): a per-size choice.
                const double sv = (a.ablate & 1) ? lr + hr
                                  : (A0 == 1 ? mag067_libm(lr, li) + mag067_libm(hr, hm) : mag067(lr, li, pow_tab) + mag067(hr, hm, pow_tab));
                if (tid + r * T < NPAIRS) {
                    buf[pp[r]] = {sv, 0.0};
                    buf[qq[r]] = {sv, 0.0};
                }
            }
        }
        __syncthreads();
        if (!(a.ablate & 4)) pfa_dft<A0, true>(buf, cs31, cs11, tid);
        double lag[4];   // Mh <= 1022 lags, four per thread: through registers, because yv lies inside buf
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = tid + r * T;
            lag[r] = 0.0;
            if (n < Mh) {
                // the real-input transform keeps residues 0..15 of axis 31; lag n at one of the others is read from lag
                // N - n (S is real and even, so is its transform)
// (A0 = 1, N = 1023: measured SLOWER with the positions kept -- 4.98-5.04 against 4.00-4.04 ms per 176 k frames, same box,
//  profiles/r6/pfa_ab.txt -- so that instantiation reads them here as before; A0 = 2: 4.56-4.69 against 4.66-4.79)
#ifndef PFA_KEEP_POS_A1
#define PFA_KEEP_POS_A1 0
#endif
                constexpr bool KEEP = A0 == 2 || PFA_KEEP_POS_A1;
                const int p0 = KEEP ? plag[r] : (int)pos[n], p1 = KEEP ? pmir[r] : (int)pos[N - n > N - 1 ? 0 : N - n];   // (lag 0: p0 % 31 <= 15 -- position 0 -- so its mirror is never read)
                lag[r] = buf[p0 % 31 > 15 ? p1 : p0].x * inv_n;
            }
        }
        __syncthreads();   // buf is dead from here on: the peak-picking scratch and yv alias it
        // the lags leave the registers enhanced (esacf.py:108-129, truncation regime: enhance_truncate) -- until round 6 they went
        // to yv raw, a barrier, and a second pass over yv did the rounds
        double* yrow = a.y_out + f * (long long)Mh;
        const int cut2 = enhance_cut2(Mh, a.enhance_mode);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = tid + r * T;
            if (n < Mh) {
                if (a.sacf_out) a.sacf_out[f * (long long)Mh + n] = lag[r];
                const double v = a.defer_enhance ? lag[r] : enhance_truncate(lag[r], n, a.n_peaks_elim, cut2);
                yv[n] = v;
                yrow[n] = v;
            }
        }
        if (!a.defer_enhance) {
            __syncthreads();
            if (!(a.ablate & (2 | 16))) {
                // (Round 6, measured and rejected: peak picking by ONE wave -- the other three finished, s_barrier counts the
                //  surviving waves only, the phases ordered by the wave's own LDS queue: 4.82-5.09 against 4.61-4.74 ms at N = 2046,
                //  3.97 against 4.01 ms at N = 1023, same bits; profiles/r6/pfa_pick_one_wave_rejected.txt.)
                peak_pick<T>(a, f, yv, smem, tid);
            }
        }
        __syncthreads();   // the next frame overwrites buf (peak-picking scratch) and yv
    }
}

// Frames whose Bluestein length exceeds the 4096 points of the in-place engine (non-power-of-two N above 2048:
// e.g. the reference's default 46.4 ms frame at 48 kHz, N = 2227 -> L = 8192): the same pipeline on the
// padded Stockham engine of mpx_fft.hpp, one 512-thread workgroup per frame and per CU (139 KB of LDS).
// Correct and complete, not tuned: ~4x the time per frame of the in-place kernel.
template <int L, int T>
__device__ __forceinline__ void dft_n_stockham(cx<double>* buf, const SacfArgs& a, cx<double>* regs, int tid) {
    const int N = a.N;
    for (int n = tid; n < L; n += T) buf[lds_slot(n)] = n < N ? cmulc(buf[lds_slot(n)], a.chirp[n]) : cx<double>{0.0, 0.0};
    __syncthreads();
    fft_lds<L, T, false, double>(buf, a.tw, regs, tid);
    // multiply by the filter spectrum; swap re/im so that the next forward FFT is an inverse one
    for (int k = tid; k < L; k += T) buf[lds_slot(k)] = cswap(cmul(buf[lds_slot(k)], a.bhat[k]));
    __syncthreads();
    fft_lds<L, T, false, double>(buf, a.tw, regs, tid);
    for (int k = tid; k < N; k += T) buf[lds_slot(k)] = cmulc(cswap(buf[lds_slot(k)]), a.chirp[k]);
    __syncthreads();
}

template <int L, int T>
__global__ __launch_bounds__(T) void sacf_big_kernel(SacfArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = a.N, Mh = a.Mh;
    cx<double>* buf = reinterpret_cast<cx<double>*>(smem);                                  // L complex, padded
    double* yv = reinterpret_cast<double*>(smem + sizeof(cx<double>) * lds_slots(L));       // Mh + 2 doubles
    const int tid = threadIdx.x;
    MPX_POW067_LDS(a, tid, T);
    const long long f = blockIdx.x;
    cx<double> regs[L / T];
    const cx<double>* xin = a.xb + band_index(f, 0, N);
    for (int n = tid; n < N; n += T) buf[lds_slot(n)] = xin[(size_t)(n >> 4) * (64 * BS_TILE) + (n & 15)];
    __syncthreads();
    dft_n_stockham<L, T>(buf, a, regs, tid);
    constexpr int PER = 4096 / T;  // N <= 4096 bins, this thread's share
    double sv[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        const int k = tid + e * T;
        sv[e] = 0.0;
        if (k < N) {
            const cx<double> A = buf[lds_slot(k)];
            const cx<double> B = cconj(buf[lds_slot(k == 0 ? 0 : N - k)]);
            const double lr = 0.5 * (A.x + B.x), li = 0.5 * (A.y + B.y);
            const double hr = 0.5 * (A.y - B.y), hm = -0.5 * (A.x - B.x);
            sv[e] = mag067(lr, li, pow_tab) + mag067(hr, hm, pow_tab);
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        const int k = tid + e * T;
        if (k < N) buf[lds_slot(k)] = {sv[e], 0.0};
    }
    __syncthreads();
    dft_n_stockham<L, T>(buf, a, regs, tid);
    const double inv_n = 1.0 / (double)N;
    for (int n = tid; n < Mh; n += T) {
        const double v = buf[lds_slot(n)].x * inv_n;
        yv[n] = v;
        if (a.sacf_out) a.sacf_out[f * (long long)Mh + n] = v;
    }
    __syncthreads();  // buf is dead from here on; the peak-picking scratch aliases it
    double* yrow = a.y_out + f * (long long)Mh;
    if (a.defer_enhance) {
        for (int n = tid; n < Mh; n += T) yrow[n] = yv[n];
        return;
    }
    {   // the truncation regime in one round (enhance_truncate)
        const int cut2 = enhance_cut2(Mh, a.enhance_mode);
        for (int n = tid; n < Mh; n += T) yv[n] = enhance_truncate(yv[n], n, a.n_peaks_elim, cut2);
        __syncthreads();
    }
    for (int n = tid; n < Mh; n += T) yrow[n] = yv[n];
    peak_pick<T>(a, f, yv, smem, tid);
}

// Even frame lengths above 4096 samples (the reference's 46.4 ms frame above 88.2 kHz: 4454 samples at 96 kHz, 8184 at
// 176.4 kHz; esacf.py:27 puts no bound on it): a 2N-1 point chirp-z no longer fits LDS, so the N-point transforms are split
// once by radix 2 around n1 = N/2 point chirp-z transforms of length L = 8192 (2 n1 - 1 <= L):
//   forward  Z = DFT_N(x_lo + i x_hi):  E = DFT_n1(z[2m]), O = DFT_n1(z[2m+1]);  Z[k] = E[k] + W_N^k O[k],
//            Z[n1 + k] = E[k] - W_N^k O[k]   (decimation in time; E waits in registers while O is computed)
//   inverse  r[n] = 1/N sum_k S[k] W_N^{-kn}:  c[j] = S[2j] + i S[2j+1] (two REAL sequences in one transform),
//            C = IDFT_n1(c);  G0[n] = (C[n] + conj C[n1-n]) / 2,  G1[n] = (C[n] - conj C[n1-n]) / 2i,
//            r[n] = (G0[n] + W_N^{-n} G1[n]) / N   (decimation in frequency of the input; n < Mh < n1)
// Three chirp-z transforms per frame instead of two; same padded Stockham engine and the same "correct and complete, not
// tuned" standing as sacf_big_kernel.  The SACF row and the peak-picking scratch alias the transform buffer (136 KB).
template <int L, int T>
__device__ __forceinline__ void chirpz_stockham(cx<double>* buf, int n1, const SacfArgs& a, cx<double>* regs, int tid) {
    for (int n = tid; n < L; n += T) buf[lds_slot(n)] = n < n1 ? cmulc(buf[lds_slot(n)], a.chirp[n]) : cx<double>{0.0, 0.0};
    __syncthreads();
    fft_lds<L, T, false, double>(buf, a.tw, regs, tid);
    for (int k = tid; k < L; k += T) buf[lds_slot(k)] = cswap(cmul(buf[lds_slot(k)], a.bhat[k]));
    __syncthreads();
    fft_lds<L, T, false, double>(buf, a.tw, regs, tid);
    for (int k = tid; k < n1; k += T) buf[lds_slot(k)] = cmulc(cswap(buf[lds_slot(k)]), a.chirp[k]);
    __syncthreads();
}

template <int L, int T>
__global__ __launch_bounds__(T) void sacf_split_kernel(SacfArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(L == 8192 && T == 512, "register shares below are written for 8192 / 512");
    const int N = a.N, Mh = a.Mh, n1 = N >> 1;
    cx<double>* buf = reinterpret_cast<cx<double>*>(smem);   // L complex, padded
    const int tid = threadIdx.x;
    MPX_POW067_LDS(a, tid, T);
    const long long f = blockIdx.x;
    cx<double> regs[L / T];
    constexpr int PER = 4096 / T;   // n1 <= 4096 points, this thread's share
    cx<double> ev[PER];
    const cx<double>* xin = a.xb + band_index(f, 0, N);
    auto sample = [&](int n) { return xin[(size_t)(n >> 4) * (64 * BS_TILE) + (n & 15)]; };
    for (int m = tid; m < n1; m += T) buf[lds_slot(m)] = sample(2 * m);
    __syncthreads();
    chirpz_stockham<L, T>(buf, n1, a, regs, tid);
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        const int k = tid + e * T;
        ev[e] = k < n1 ? buf[lds_slot(k)] : cx<double>{0.0, 0.0};
    }
    __syncthreads();
    for (int m = tid; m < n1; m += T) buf[lds_slot(m)] = sample(2 * m + 1);
    __syncthreads();
    chirpz_stockham<L, T>(buf, n1, a, regs, tid);
    cx<double>* E = buf + lds_slot(L / 2);   // upper half: E[k] at E[lds_slot(k)] (lds_slot(L/2 + k) = lds_slot(L/2) + lds_slot(k))
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        const int k = tid + e * T;
        if (k < n1) E[lds_slot(k)] = ev[e];
    }
    __syncthreads();
    // S[k] = |X_lo[k]|^0.67 + |X_hi[k]|^0.67 for k = 0 .. n1 (S[N-k] = S[k]); thread t takes k = t + 512 e, and k = n1 last
    double sv[PER + 1];
#pragma unroll
    for (int e = 0; e <= PER; ++e) {
        const int k = tid + e * T;
        sv[e] = 0.0;
        if (k <= n1) {
            const int ka = k < n1 ? k : 0, kb = k == 0 ? 0 : n1 - k;   // Z[k] from (E, O)[ka]; Z[N-k] from (E, O)[kb]
            const cx<double> wa = a.twn[ka], wb = a.twn[kb];
            const cx<double> Ea = E[lds_slot(ka)], Oa = cmul(buf[lds_slot(ka)], wa);
            const cx<double> Eb = E[lds_slot(kb)], Ob = cmul(buf[lds_slot(kb)], wb);
            const double sa = k < n1 ? 1.0 : -1.0, sb = k == 0 ? 1.0 : -1.0;
            const cx<double> A = {Ea.x + sa * Oa.x, Ea.y + sa * Oa.y};
            const cx<double> B = {Eb.x + sb * Ob.x, -(Eb.y + sb * Ob.y)};   // conj Z[N-k]
            const double lr = 0.5 * (A.x + B.x), li = 0.5 * (A.y + B.y);
            const double hr = 0.5 * (A.y - B.y), hm = -0.5 * (A.x - B.x);
            sv[e] = mag067(lr, li, pow_tab) + mag067(hr, hm, pow_tab);
        }
    }
    __syncthreads();
    double* sd = reinterpret_cast<double*>(E);   // S[0 .. n1], read back folded
#pragma unroll
    for (int e = 0; e <= PER; ++e) {
        const int k = tid + e * T;
        if (k <= n1) sd[k] = sv[e];
    }
    __syncthreads();
    for (int j = tid; j < n1; j += T) {
        const int k0 = 2 * j, k1 = 2 * j + 1;
        // swapped: the forward chirp-z of (im, re) is the swapped inverse transform
        buf[lds_slot(j)] = {sd[k1 <= n1 ? k1 : N - k1], sd[k0 <= n1 ? k0 : N - k0]};
    }
    __syncthreads();
    chirpz_stockham<L, T>(buf, n1, a, regs, tid);
    const double inv_n = 1.0 / (double)N;
    double rv[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        const int n = tid + e * T;
        rv[e] = 0.0;
        if (n < Mh) {
            const cx<double> C = cswap(buf[lds_slot(n)]);
            const cx<double> Cc = cconj(cswap(buf[lds_slot(n == 0 ? 0 : n1 - n)]));
            const cx<double> w = a.twn[n];                                      // W_N^n; its conjugate rotates G1
            const double g0 = 0.5 * (C.x + Cc.x);
            const double qx = 0.5 * (C.y - Cc.y), qy = -0.5 * (C.x - Cc.x);    // G1[n]
            rv[e] = (g0 + (w.x * qx + w.y * qy)) * inv_n;
        }
    }
    __syncthreads();   // buf is dead from here on: the row and the peak-picking scratch alias it
    double* yv = reinterpret_cast<double*>(smem + peak_scratch_bytes(Mh));
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        const int n = tid + e * T;
        if (n < Mh) {
            yv[n] = rv[e];
            if (a.sacf_out) a.sacf_out[f * (long long)Mh + n] = rv[e];
        }
    }
    __syncthreads();
    double* yrow = a.y_out + f * (long long)Mh;
    if (a.defer_enhance) {
        for (int n = tid; n < Mh; n += T) yrow[n] = yv[n];
        return;
    }
    {   // the truncation regime in one round (enhance_truncate)
        const int cut2 = enhance_cut2(Mh, a.enhance_mode);
        for (int n = tid; n < Mh; n += T) yv[n] = enhance_truncate(yv[n], n, a.n_peaks_elim, cut2);
        __syncthreads();
    }
    for (int n = tid; n < Mh; n += T) yrow[n] = yv[n];
    peak_pick<T>(a, f, yv, smem, tid);
}

// Frame lengths no LDS engine holds: odd above 4096 samples, anything above 8192 (esacf.py:27 takes int(fs * 46.4 / 1000)
// whatever fs is: 192 kHz -> 8908 samples).  The N-point transforms are chirp-z convolutions of L = 8192 R points, R = 2 or 4
// (2N - 1 <= L), and an L-point transform is R transforms of 8192 points on the padded Stockham engine around one radix-R
// step that never leaves the workgroup:
//   forward, decimation in frequency:  A[R k + r] = FFT_8192(u_r)[k],  u_r[m] = W_L^{m r} sum_j a[m + 8192 j] W_R^{j r}
//            (a = z conj(chirp), zero from N on: j < R / 2 only)
//   times the filter spectrum, stored as [r][k]
//   inverse, decimation in time:       y[n] = sum_r conj(W_L^{n r}) IFFT_8192(Y_r)[n mod 8192]
// so each residue r is transformed, filtered and transformed back while it is in LDS, and only the wanted outputs (N bins,
// then (N - 1) / 2 lags) are accumulated, in a per-workgroup scratch row in HBM (24 N bytes, L2 / MALL resident: the grid is
// one persistent workgroup per CU).  4 R transforms of 8192 points per frame.  Correct and complete, not tuned.
struct HugeArgs {
    const cx<double>* twL;     // [L] W_L^j
    const cx<double>* bhat_r;  // [R][8192] FFT_L(chirp filter) / L at bins R k + r
    cx<double>* scratch;       // [grid][2 N]: N accumulators, then N doubles of |X|^0.67 sums
    int R, L;
};

template <int T, typename Src>
__device__ __forceinline__ void chirpz_huge(cx<double>* buf, const SacfArgs& a, const HugeArgs& h, int nout, Src src,
                                            cx<double>* acc, cx<double>* regs, int tid) {
    constexpr int M = 8192;
    const int N = a.N, R = h.R, Lm = h.L - 1;
    for (int rp = 0; rp < R; ++rp) {
        for (int m = tid; m < M; m += T) {
            cx<double> u = {0.0, 0.0};
            if (m < N) u = cmulc(src(m), a.chirp[m]);
            if (R == 4 && m + M < N) {
                cx<double> v = cmulc(src(m + M), a.chirp[m + M]);   // times W_4^rp = (-i)^rp
                if (rp == 1) v = {v.y, -v.x};
                else if (rp == 2) v = {-v.x, -v.y};
                else if (rp == 3) v = {-v.y, v.x};
                u = cadd(u, v);
            }
            buf[lds_slot(m)] = cmul(u, h.twL[(m * rp) & Lm]);
        }
        __syncthreads();
        fft_lds<M, T, false, double>(buf, a.tw, regs, tid);
        const cx<double>* bh = h.bhat_r + (size_t)rp * M;
        // swapped: the next forward transform is then the inverse one
        for (int k = tid; k < M; k += T) buf[lds_slot(k)] = cswap(cmul(buf[lds_slot(k)], bh[k]));
        __syncthreads();
        fft_lds<M, T, false, double>(buf, a.tw, regs, tid);
        for (int n = tid; n < nout; n += T) {   // a thread owns its outputs across the residues: no barrier for acc
            const cx<double> t = cmulc(cswap(buf[lds_slot(n & (M - 1))]), h.twL[(n * rp) & Lm]);
            acc[n] = rp == 0 ? t : cadd(acc[n], t);
        }
        __syncthreads();
    }
    for (int n = tid; n < nout; n += T) acc[n] = cmulc(acc[n], a.chirp[n]);
    __syncthreads();
}

template <int T>
__global__ __launch_bounds__(T) void sacf_huge_kernel(SacfArgs a, HugeArgs h) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int M = 8192;
    const int N = a.N, Mh = a.Mh;
    cx<double>* buf = reinterpret_cast<cx<double>*>(smem);   // 8192 complex, padded
    const int tid = threadIdx.x;
    MPX_POW067_LDS(a, tid, T);
    cx<double> regs[M / T];
    cx<double>* acc = h.scratch + (size_t)blockIdx.x * 2 * N;
    double* sreal = reinterpret_cast<double*>(acc + N);
    const double inv_n = 1.0 / (double)N;
    for (long long f = blockIdx.x; f < a.num_frames; f += gridDim.x) {
        const cx<double>* xin = a.xb + band_index(f, 0, N);
        chirpz_huge<T>(buf, a, h, N, [&](int n) { return xin[(size_t)(n >> 4) * (64 * BS_TILE) + (n & 15)]; }, acc, regs, tid);
        for (int k = tid; k < N; k += T) {
            const cx<double> A = acc[k];
            const cx<double> B = cconj(acc[k == 0 ? 0 : N - k]);
            const double lr = 0.5 * (A.x + B.x), li = 0.5 * (A.y + B.y);
            const double hr = 0.5 * (A.y - B.y), hm = -0.5 * (A.x - B.x);
            sreal[k] = mag067(lr, li, pow_tab) + mag067(hr, hm, pow_tab);
        }
        __syncthreads();
        // S is real and symmetric: its inverse transform is its forward transform (over N)
        chirpz_huge<T>(buf, a, h, Mh, [&](int n) { return cx<double>{sreal[n], 0.0}; }, acc, regs, tid);
        for (int n = tid; n < Mh; n += T) {
            const double v = acc[n].x * inv_n;
            a.y_out[f * (long long)Mh + n] = v;
            if (a.sacf_out) a.sacf_out[f * (long long)Mh + n] = v;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ kernel 2b / 2c
// Enhancement when librosa.effects.time_stretch is a REAL phase vocoder, i.e. when the STFT of the
// Mh-lag SACF has more than two frames (Mh >= 1024: ESACF frames above 2048 samples).  Per rate r
// (esacf.py:117-127, librosa >= 0.10 semantics as restated in oracle/thirdparty.py):
//   clip -> STFT (n_fft 2048, hop 512, periodic Hann, centered, zero padded) of the columns the
//   vocoder reads -> phase_vocoder (integer rate => alpha = 0; <= 2 output frames since n_frames <= 4)
//   -> ISTFT (overlap-add / window sum-square, drop 1024, length round(Mh/r)) -> subtract -> clip.
// Two real frames share one complex 2048-point LDS FFT in both directions (re/im packing).
constexpr int PV_T = 256, PV_NFFT = 2048, PV_HOP = 512, PV_BINS = PV_NFFT / 2 + 1;
#ifndef PV_WGS
#define PV_WGS 3
#endif

struct PvArgs {
    double* y;                 // [F, Mh] SACF in, enhanced SACF out
    int Mh;
    int n_peaks_elim;
    const cx<double>* tw;      // W_2048
};

__device__ __forceinline__ double pv_hann(const cx<double>* __restrict__ tw, int n) {
    return 0.5 - 0.5 * tw[n & (PV_NFFT - 1)].x;  // periodic Hann: cos(2 pi n / 2048) = Re W_2048^n
}

// PICK: the peak picking of the finished row runs at the end of this kernel, on the LDS copy it already holds (until round 3:
// a second kernel that read the row back from HBM).
// MAXS: output frames of the vocoder this instantiation can hold (2: rows up to 2047 lags, everything until round 3;
// 4: rows up to 4095 lags, ESACF frames up to 8192 samples -- the phase is then carried as a unit vector per bin, see there).
template <bool PICK, int MAXS>
__global__ __launch_bounds__(PV_T, MAXS > 2 ? 1 : PV_WGS) void pv_enhance_kernel(PvArgs a, SacfArgs sa) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cx<double>* buf = reinterpret_cast<cx<double>*>(smem);                       // 2048 complex (padded)
    double* x = reinterpret_cast<double*>(buf + lds_slots(PV_NFFT));             // [Mh] working copy of the SACF
    const int tid = threadIdx.x, Mh = a.Mh;
    const long long f = blockIdx.x;
    double* row = a.y + f * (long long)Mh;
    cx<double> regs[PV_NFFT / PV_T];
    // A thread always owns the rfft bins k = tid + 256 j (j < 5; bin 1024 is thread 0's fifth): the phase
    // accumulator and the synthesis spectra of both output frames stay in its registers, and it writes bin k AND
    // its Hermitian mirror 2048-k of the packed inverse transform itself.  LDS is the FFT buffer + the row: 51 KB,
    // two workgroups per CU (the spectra in LDS made it 92 KB and one).
    constexpr int NB = (PV_BINS + PV_T - 1) / PV_T;  // 5
    cx<double> d0[NB], d1[NB];
    cx<double> d2[MAXS > 2 ? NB : 1], d3[MAXS > 2 ? NB : 1], dir[MAXS > 2 ? NB : 1];
    for (int n = tid; n < Mh; n += PV_T) x[n] = row[n];
    __syncthreads();
    const int n_frames = 1 + Mh / PV_HOP;  // centered STFT of Mh samples
    for (int r = 2; r <= a.n_peaks_elim; ++r) {
        for (int n = tid; n < Mh; n += PV_T) x[n] = x[n] < 0.0 ? 0.0 : x[n];  // clip
        __syncthreads();
        const int len_out = (int)nearbyint((double)Mh / (double)r);          // int(round(len/rate)), half-to-even
        const int nsteps = (n_frames + r - 1) / r;                            // len(arange(0, n_frames, r)) <= 2
        if (nsteps < 2) {
            // ONE output frame: the vocoder passes STFT column 0 through unchanged, and the ISTFT of a single frame
            // divides out the window it was analysed with -- time_stretch returns x[:len_out] itself (to the rounding of
            // its own two FFTs, 1e-16 of the row maximum; the window is >= 0.25 over these samples), so the subtraction
            // leaves zeros there.  The same identity sacf_kernel uses for rows below 1024 lags; two 2048-point
            // transforms per rate saved (rates 4, 5, 6 of the default six at Mh = 2047: half of this kernel's transforms).
            for (int i = tid; i < len_out && i < Mh; i += PV_T) x[i] = 0.0;
            __syncthreads();
            continue;
        }
#pragma unroll
        for (int t = 0; t < MAXS; ++t) {
            if (t >= nsteps) break;
            const int c0 = t * r, c1 = c0 + 1;                                // STFT columns int(step), int(step)+1
            // analysis: frame c covers xpad[c*512 + n], xpad = [1024 zeros | x | 1024 zeros]
            for (int n = tid; n < PV_NFFT; n += PV_T) {
                const int i0 = c0 * PV_HOP + n - PV_NFFT / 2, i1 = c1 * PV_HOP + n - PV_NFFT / 2;
                const double w = pv_hann(a.tw, n);
                const double v0 = (i0 >= 0 && i0 < Mh) ? x[i0] * w : 0.0;
                const double v1 = (c1 < n_frames && i1 >= 0 && i1 < Mh) ? x[i1] * w : 0.0;
                buf[lds_slot(n)] = {v0, v1};
            }
            __syncthreads();
            fft_lds<PV_NFFT, PV_T, false, double>(buf, a.tw, regs, tid);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int k = tid + j * PV_T;
                if (k < PV_BINS) {
                    const cx<double> Z = buf[lds_slot(k & (PV_NFFT - 1))];
                    const cx<double> Zc = cconj(buf[lds_slot((PV_NFFT - k) & (PV_NFFT - 1))]);
                    const cx<double> A = {0.5 * (Z.x + Zc.x), 0.5 * (Z.y + Zc.y)};   // rfft of column c0
                    const cx<double> B = {0.5 * (Z.y - Zc.y), -0.5 * (Z.x - Zc.x)};  // rfft of column c1
                    // librosa: D'[:, t] = |column c0| e^{i phase_acc}, phase_acc = angle(D[:, 0]) at t = 0 and then
                    // phase_acc += phi + wrap(angle(c1) - angle(c0) - phi).  With at most two output frames this needs
                    // no angle at all: at t = 0, |A| e^{i angle(A)} is A itself, and the accumulated phase of t = 1 is
                    // angle(column 1 of step 0) up to a multiple of 2 pi, so D'[:, 1] = |column r| * B_0 / |B_0|
                    // (angle(0) = 0 -> direction (1, 0)).  That is the reference's value to the ~2e-13 rad its own
                    // phi = pi*512*k/1024 arithmetic carries, and replaces two atan2, a hypot, a cos and a sin per
                    // bin and step -- most of this kernel's instructions -- by a square root and a division.
                    if (t == 0) {
                        d0[j] = A;
                        const double nb = sqrt(B.x * B.x + B.y * B.y);
                        const double inb = nb > 0.0 ? 1.0 / nb : 0.0;
                        d1[j] = nb > 0.0 ? cx<double>{B.x * inb, B.y * inb} : cx<double>{1.0, 0.0};   // direction for t = 1
                    } else if (MAXS <= 2) {
                        const double mag = sqrt(A.x * A.x + A.y * A.y);
                        d1[j] = {mag * d1[j].x, mag * d1[j].y};
                    } else {
                        // more than two output frames: e^{i phase_acc} of frame t+1 is that of frame t turned by
                        // angle(c1) - angle(c0) (phi + wrap(dphase - phi) is dphase up to a multiple of 2 pi), i.e. times
                        // u(B) conj(u(A)) with u(z) = z / |z| and u(0) = 1 (np.angle(0) = 0): still no angle, one more
                        // square root and division per bin and step
                        const cx<double> dcur = t == 1 ? d1[j] : dir[j];
                        const double mag = sqrt(A.x * A.x + A.y * A.y);
                        const cx<double> val = {mag * dcur.x, mag * dcur.y};
                        if (t == 1) d1[j] = val;
                        else if (t == 2) d2[j] = val;
                        else d3[j] = val;
                        if (t + 1 < nsteps) {
                            const double ia = mag > 0.0 ? 1.0 / mag : 0.0;
                            const cx<double> ua = mag > 0.0 ? cx<double>{A.x * ia, A.y * ia} : cx<double>{1.0, 0.0};
                            const double nb = sqrt(B.x * B.x + B.y * B.y);
                            const double inb = nb > 0.0 ? 1.0 / nb : 0.0;
                            const cx<double> ub = nb > 0.0 ? cx<double>{B.x * inb, B.y * inb} : cx<double>{1.0, 0.0};
                            const cx<double> turn = cmulc(ub, ua);   // ub * conj(ua)
                            dir[j] = cmul(dcur, turn);
                        }
                    }
                }
            }
            __syncthreads();
        }
        if constexpr (MAXS <= 2) {
            // synthesis: irfft of both output frames through one complex inverse FFT (swap trick); every thread
            // writes its bins and their Hermitian mirrors
    #pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int k = tid + j * PV_T;
                if (k < PV_BINS) {
                    cx<double> e0 = d0[j], e1 = d1[j];
                    if (k == 0 || k == PV_NFFT / 2) e0.y = e1.y = 0.0;    // irfft ignores the imaginary part there
                    buf[lds_slot(k & (PV_NFFT - 1))] = cswap(cx<double>{e0.x - e1.y, e0.y + e1.x});  // D0 + i*D1
                    if (k != 0 && k != PV_NFFT / 2)                         // conj(D0) + i*conj(D1) at 2048-k
                        buf[lds_slot(PV_NFFT - k)] = cswap(cx<double>{e0.x + e1.y, -e0.y + e1.x});
                }
            }
            __syncthreads();
            fft_lds<PV_NFFT, PV_T, false, double>(buf, a.tw, regs, tid);
            // overlap-add, normalise by the window sum-square, drop the 1024-sample centre pad, subtract, clip
            const double inv = 1.0 / (double)PV_NFFT;
            for (int i = tid; i < Mh; i += PV_T) {
                double v = x[i];
                if (i < len_out) {
                    const int n = i + PV_NFFT / 2;  // position in the overlap-add buffer
                    double acc = 0.0, wss = 0.0;
                    if (n < PV_NFFT) {
                        const double w = pv_hann(a.tw, n);
                        acc += w * (cswap(buf[lds_slot(n)]).x * inv);
                        wss += w * w;
                    }
                    if (nsteps > 1 && n >= PV_HOP && n - PV_HOP < PV_NFFT) {
                        const double w = pv_hann(a.tw, n - PV_HOP);
                        acc += w * (cswap(buf[lds_slot(n - PV_HOP)]).y * inv);
                        wss += w * w;
                    }
                    if (wss > 2.2250738585072014e-308) acc /= wss;
                    v -= acc;
                }
                x[i] = v < 0.0 ? 0.0 : v;
            }
            __syncthreads();
        } else {
            // up to four output frames: two packed inverse transforms; the overlap-add of a thread's samples
            // (i = tid + 256 q < len_out <= 2048) is carried in registers between them, frames in ascending order
            constexpr int NQ = PV_NFFT / PV_T;
            const double inv = 1.0 / (double)PV_NFFT;
            double acc[NQ], wss[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[q] = wss[q] = 0.0;
#pragma unroll
            for (int pr = 0; pr < MAXS / 2; ++pr) {
                if (2 * pr >= nsteps) break;
                const bool second = 2 * pr + 1 < nsteps;
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int k = tid + j * PV_T;
                    if (k < PV_BINS) {
                        cx<double> e0 = pr == 0 ? d0[j] : d2[j], e1 = pr == 0 ? d1[j] : d3[j];
                        if (!second) e1 = {0.0, 0.0};
                        if (k == 0 || k == PV_NFFT / 2) e0.y = e1.y = 0.0;
                        buf[lds_slot(k & (PV_NFFT - 1))] = cswap(cx<double>{e0.x - e1.y, e0.y + e1.x});
                        if (k != 0 && k != PV_NFFT / 2)
                            buf[lds_slot(PV_NFFT - k)] = cswap(cx<double>{e0.x + e1.y, -e0.y + e1.x});
                    }
                }
                __syncthreads();
                fft_lds<PV_NFFT, PV_T, false, double>(buf, a.tw, regs, tid);
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int i = tid + q * PV_T;
                    if (i < len_out) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int m = i + PV_NFFT / 2 - (2 * pr + h) * PV_HOP;   // sample of output frame 2 pr + h
                            if ((h == 0 || second) && m >= 0 && m < PV_NFFT) {
                                const double w = pv_hann(a.tw, m);
                                const cx<double> z = cswap(buf[lds_slot(m)]);
                                acc[q] += w * ((h == 0 ? z.x : z.y) * inv);
                                wss[q] += w * w;
                            }
                        }
                    }
                }
                __syncthreads();
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int i = tid + q * PV_T;
                if (i < len_out && i < Mh) {
                    double t = acc[q];
                    if (wss[q] > 2.2250738585072014e-308) t /= wss[q];
                    const double v = x[i] - t;
                    x[i] = v < 0.0 ? 0.0 : v;
                }
            }
            __syncthreads();
        }
    }
    for (int n = tid; n < Mh; n += PV_T) row[n] = x[n];
    if (PICK) {
        __syncthreads();   // the transform buffer is dead: it is the picker's scratch
        peak_pick<PV_T>(sa, f, x, reinterpret_cast<char*>(buf), tid);
    }
}

// The same enhancement for rows of 4096 ... 8191 lags (sacf_huge_kernel's): up to MAXS = 8 output frames per rate, carried
// as in pv_enhance_kernel<., 4> (unit vectors instead of angles), the synthesis spectra of all output frames in registers
// (one workgroup per CU, a wave per SIMD: 512 registers), the row in LDS (64 KB), peak picking on 128 flag words.
template <int MAXS>
__global__ __launch_bounds__(PV_T, 1) void pv_enhance_big_kernel(PvArgs a, SacfArgs sa) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, Mh = a.Mh;
    size_t front = sizeof(cx<double>) * lds_slots(PV_NFFT);
    if (peak_scratch_bytes(Mh) > front) front = peak_scratch_bytes(Mh);
    cx<double>* buf = reinterpret_cast<cx<double>*>(smem);      // 2048 complex (padded); later the picker's scratch
    double* x = reinterpret_cast<double*>(smem + front);        // [Mh] working copy of the SACF
    const long long f = blockIdx.x;
    double* row = a.y + f * (long long)Mh;
    cx<double> regs[PV_NFFT / PV_T];
    constexpr int NB = (PV_BINS + PV_T - 1) / PV_T;  // 5
    constexpr int NQ = MAXS * PV_HOP / PV_T;         // len_out <= MAXS * 512 samples, a thread's share
    cx<double> d[MAXS][NB], dir[NB];
    for (int n = tid; n < Mh; n += PV_T) x[n] = row[n];
    __syncthreads();
    const int n_frames = 1 + Mh / PV_HOP;
    for (int r = 2; r <= a.n_peaks_elim; ++r) {
        for (int n = tid; n < Mh; n += PV_T) x[n] = x[n] < 0.0 ? 0.0 : x[n];
        __syncthreads();
        const int len_out = (int)nearbyint((double)Mh / (double)r);
        const int nsteps = (n_frames + r - 1) / r;
        if (nsteps < 2) {   // one output frame: time_stretch returns x[:len_out] (see pv_enhance_kernel)
            for (int i = tid; i < len_out && i < Mh; i += PV_T) x[i] = 0.0;
            __syncthreads();
            continue;
        }
#pragma unroll
        for (int t = 0; t < MAXS; ++t) {
            if (t >= nsteps) break;
            const int c0 = t * r, c1 = c0 + 1;
            for (int n = tid; n < PV_NFFT; n += PV_T) {
                const int i0 = c0 * PV_HOP + n - PV_NFFT / 2, i1 = c1 * PV_HOP + n - PV_NFFT / 2;
                const double w = pv_hann(a.tw, n);
                const double v0 = (i0 >= 0 && i0 < Mh) ? x[i0] * w : 0.0;
                const double v1 = (c1 < n_frames && i1 >= 0 && i1 < Mh) ? x[i1] * w : 0.0;
                buf[lds_slot(n)] = {v0, v1};
            }
            __syncthreads();
            fft_lds<PV_NFFT, PV_T, false, double>(buf, a.tw, regs, tid);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int k = tid + j * PV_T;
                if (k < PV_BINS) {
                    const cx<double> Z = buf[lds_slot(k & (PV_NFFT - 1))];
                    const cx<double> Zc = cconj(buf[lds_slot((PV_NFFT - k) & (PV_NFFT - 1))]);
                    const cx<double> A = {0.5 * (Z.x + Zc.x), 0.5 * (Z.y + Zc.y)};   // rfft of column c0
                    const cx<double> B = {0.5 * (Z.y - Zc.y), -0.5 * (Z.x - Zc.x)};  // rfft of column c1
                    const double nb = sqrt(B.x * B.x + B.y * B.y);
                    const double inb = nb > 0.0 ? 1.0 / nb : 0.0;
                    const cx<double> ub = nb > 0.0 ? cx<double>{B.x * inb, B.y * inb} : cx<double>{1.0, 0.0};
                    if (t == 0) {
                        d[0][j] = A;        // |A| e^{i angle(A)}
                        dir[j] = ub;        // e^{i phase_acc} of frame 1: angle(A) + (angle(B) - angle(A))
                    } else {
                        const double mag = sqrt(A.x * A.x + A.y * A.y);
                        d[t][j] = {mag * dir[j].x, mag * dir[j].y};
                        if (t + 1 < nsteps) {
                            const double ia = mag > 0.0 ? 1.0 / mag : 0.0;
                            const cx<double> ua = mag > 0.0 ? cx<double>{A.x * ia, A.y * ia} : cx<double>{1.0, 0.0};
                            dir[j] = cmul(dir[j], cmulc(ub, ua));
                        }
                    }
                }
            }
            __syncthreads();
        }
        const double inv = 1.0 / (double)PV_NFFT;
        double acc[NQ];   // overlap-add of a thread's samples i = tid + 256 q, frames in ascending order
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] = 0.0;
#pragma unroll
        for (int pr = 0; pr < MAXS / 2; ++pr) {
            if (2 * pr >= nsteps) break;
            const bool second = 2 * pr + 1 < nsteps;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int k = tid + j * PV_T;
                if (k < PV_BINS) {
                    cx<double> e0 = d[2 * pr][j], e1 = d[2 * pr + 1][j];
                    if (!second) e1 = {0.0, 0.0};
                    if (k == 0 || k == PV_NFFT / 2) e0.y = e1.y = 0.0;
                    buf[lds_slot(k & (PV_NFFT - 1))] = cswap(cx<double>{e0.x - e1.y, e0.y + e1.x});
                    if (k != 0 && k != PV_NFFT / 2)
                        buf[lds_slot(PV_NFFT - k)] = cswap(cx<double>{e0.x + e1.y, -e0.y + e1.x});
                }
            }
            __syncthreads();
            fft_lds<PV_NFFT, PV_T, false, double>(buf, a.tw, regs, tid);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int i = tid + q * PV_T;
                if (i < len_out) {
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const int m = i + PV_NFFT / 2 - (2 * pr + hh) * PV_HOP;   // sample of output frame 2 pr + hh
                        if ((hh == 0 || second) && m >= 0 && m < PV_NFFT) {
                            const double w = pv_hann(a.tw, m);
                            const cx<double> z = cswap(buf[lds_slot(m)]);
                            acc[q] += w * ((hh == 0 ? z.x : z.y) * inv);
                        }
                    }
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = tid + q * PV_T;
            if (i < len_out && i < Mh) {
                double t = acc[q], wss = 0.0;   // window sum-square over the frames that cover sample i, same order
                for (int fr = 0; fr < nsteps; ++fr) {
                    const int m = i + PV_NFFT / 2 - fr * PV_HOP;
                    if (m >= 0 && m < PV_NFFT) {
                        const double w = pv_hann(a.tw, m);
                        wss += w * w;
                    }
                }
                if (wss > 2.2250738585072014e-308) t /= wss;
                const double v = x[i] - t;
                x[i] = v < 0.0 ? 0.0 : v;
            }
        }
        __syncthreads();
    }
    for (int n = tid; n < Mh; n += PV_T) row[n] = x[n];
    __syncthreads();
    peak_pick<PV_T, true>(sa, f, x, smem, tid);
}

// Rows above 4095 lags without the phase vocoder (`noop` enhancement, or fewer than two rates): esacf.py:117-127's
// clip / subtract / clip on the raw row, then the 128-word peak picking.
template <int T>
__global__ __launch_bounds__(T) void enhance_pick_big_kernel(SacfArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int Mh = a.Mh, tid = threadIdx.x;
    const long long f = blockIdx.x;
    double* yv = reinterpret_cast<double*>(smem + peak_scratch_bytes(Mh));
    double* row = a.y_out + f * (long long)Mh;
    for (int n = tid; n < Mh; n += T) yv[n] = row[n];
    __syncthreads();
    {   // the truncation regime in one round (enhance_truncate)
        const int cut2 = enhance_cut2(Mh, a.enhance_mode);
        for (int n = tid; n < Mh; n += T) yv[n] = enhance_truncate(yv[n], n, a.n_peaks_elim, cut2);
        __syncthreads();
    }
    for (int n = tid; n < Mh; n += T) row[n] = yv[n];
    peak_pick<T, true>(a, f, yv, smem, tid);
}

// Peak picking on rows that are already enhanced (phase-vocoder regime).
template <int T>
__global__ __launch_bounds__(T) void peakpick_kernel(SacfArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int Mh = a.Mh, tid = threadIdx.x;
    const long long f = blockIdx.x;
    char* scratch = smem;
    double* yv = reinterpret_cast<double*>(smem + peak_scratch_bytes(Mh));
    const double* row = a.y_out + f * (long long)Mh;
    for (int n = tid; n < Mh; n += T) yv[n] = row[n];
    __syncthreads();
    peak_pick<T>(a, f, yv, scratch, tid);
}

// ------------------------------------------------------------------ kernel 3
// Gaussian peak fit, one lane per peak, PERSISTENT lanes with dynamic work fetch.
// MINPACK's iteration count is data dependent (typically ~50 function evaluations, but 1-2 % of the
// peaks -- runaway fits near the end of the lag range -- burn the full maxfev = 800).  With a static
// "lane w fits peak w" mapping nearly every wave contains such a straggler and waits for it (measured:
// 2.6 M VALU instructions per wave instead of ~0.1 M).  Here lm::gaussian_fit (mpx_lm.hpp) is unrolled into a
// resumable state machine: a lane that finishes its peak pulls the next one from a global counter
// (wave-aggregated atomic), so stragglers only delay themselves.
// Per trip of the main loop a lane runs at most one OUTER step (jacobian, pivoted Householder QR, Q^T f)
// followed by one INNER step (lmpar, trial point, ratio test); both are straight-line code shared by all
// lanes in that state.  Jacobian + samples are VGPRs, MINPACK's two m-vectors are LDS [row][lane].
// The only per-fit vector that must survive between trips is MINPACK's fvec (LDS, [row][lane]); the 21 samples
// are re-read from the ESACF row (L2) at the start of each step, the trial residuals and the copy of fvec
// that the reflections turn into Q^T f are registers that share the space of the (by then dead) samples /
// jacobian.  That keeps the kernel at <= 256 registers and 10.5 KB of LDS per wave: 2 waves per SIMD.
// Divisions by a common denominator (forward-difference step, Householder norm, 2 dev^2) are one reciprocal
// and multiplies: 1-2 ulp away from MINPACK's quotients, far inside what the forward-difference jacobian
// (noise ~1e-8) and the tolerances (1.5e-8) resolve.
constexpr int FIT_THREADS = 128;
constexpr int FIT_WAVES_PER_SIMD = 2;
#if defined(MPX_DEV_KNOBS) && defined(MPX_FIT_STATS)
// statistics of the fit kernels (builds with -DMPX_DEV_KNOBS -DMPX_FIT_STATS only -- `make stats` -> libmpx_hip_stats.so; the
// counters are global atomics in the trip loop and slow the kernels down thirty-fold; read with mpx_dev_fit_hist, scripts/dev/fit_hist.py):
// [0..11] lane kernel: lmpar iterations per trial (11 = the trial took no lmpar: FIT_INIT); [12] INNER sections run (waves),
// [13] lanes in them, [14] sum of the wave maximum of the lmpar iterations, [15] sum of the lanes' iterations, [16] OUTER
// sections run (waves), [17] lanes in them, [18] trips (waves); [20..30] cooperative kernels: lmpar iterations per trial and
// fit; [32] their trips (waves), [33] sum of the wave maximum, [34] fits in those trips, [35] inner-loop repeats
__device__ unsigned long long g_fit_hist[64];
#define FIT_STAT(i, v) atomicAdd(&g_fit_hist[i], (unsigned long long)(v))
#else
#define FIT_STAT(i, v) ((void)0)
#endif
enum { FIT_NEED_WORK = 0, FIT_OUTER = 1, FIT_INNER = 2, FIT_DONE = 3, FIT_INIT = 4 };

// exp(x) for x <= 0 (the gaussian's argument; NaN propagates): x = (64 e + j) ln2/64 + r, |r| <= ln2/128, so
// exp(x) = 2^e * T[j] * (1 + r + r^2/2 + ... + r^5/120) with T[j] = 2^(j/64) from a 64-entry LDS table; the
// truncated series is below 0.35 ulp at that |r| and the whole thing stays within ~1 ulp of libm's exp at
// 19 VALU instructions + one LDS read instead of ~31.  The fused multiply-adds are spelled as v_fma_f64 so
// that the compiler cannot turn them into a v_mov + v_fmac pair per coefficient (it does for libm's Horner
// chain: 9 moves per exponential).
// (one SGPR-pair constant per instruction: the constant-bus limit of a gfx9 VOP3; constants passed as VGPR operands
// got spilled to scratch and reloaded behind an s_waitcnt in front of every use)
__device__ __forceinline__ double fma_vvv(double a, double b, double c) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ double fma_vsv(double a, double k, double c) {  // a * k + c, k a constant
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(k), "v"(c));
    return d;
}
__device__ __forceinline__ double fma_vvs(double a, double b, double k) {  // a * b + k, k a constant
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(k));
    return d;
}
__device__ __forceinline__ double mul_vs(double a, double k) {
    double d;
    asm("v_mul_f64 %0, %1, %2" : "=v"(d) : "v"(a), "s"(k));
    return d;
}
__device__ __forceinline__ double add_vs(double a, double k) {
    double d;
    asm("v_add_f64 %0, %1, %2" : "=v"(d) : "v"(a), "s"(k));
    return d;
}
__device__ __forceinline__ double exp_nonpos(double x, const double* __restrict__ tab) {
    const double xc = x < -750.0 ? -750.0 : x;  // exp(-750) = 0 in fp64; keeps the integer conversion in range
    const double n = rint(mul_vs(xc, 92.33248261689366));         // 64 / ln 2
    double r = fma_vsv(n, -0.01083042469326756, xc);              // ln2/64, high part (32 significant bits)
    r = fma_vsv(n, -2.9815858269852933e-12, r);                   // ... low part
    const int ni = (int)n;
    const double t = tab[ni & 63];
    double q = add_vs(mul_vs(r, 1.0 / 120.0), 1.0 / 24.0);        // r/120 + 1/24
    q = fma_vvs(r, q, 1.0 / 6.0);
    q = fma_vvs(r, q, 0.5);
    const double p = fma_vvv(r * r, q, r);                        // e^r - 1
    return ldexp(fma_vvv(t, p, t), ni >> 6);
}

// c ? v : 0 with v evaluated unconditionally.  The 21 samples of a window are independent dependency chains
// (an exponential each); behind per-sample `if (i < m)` branches they run one after the other, as selects the
// scheduler interleaves them.  Rows >= m read stale LDS / compute on zero samples and are discarded here.
__device__ __forceinline__ double keep_if(bool c, double v) {
    asm volatile("" : "+v"(v));
    return c ? v : 0.0;
}

#pragma clang fp contract(off)  // until the end of peakfit_kernel: see below
// ---- ONE arithmetic for both fit kernels ---------------------------------------------------------------
// peakfit_kernel (a fit per lane, its 21 rows in registers) and coopfit_kernel (a fit per 16-lane row, rows l and
// l + 16 per lane) must produce the SAME BITS for the same fit: which of them finishes a runaway fit depends on the
// timing of a launch, so any difference in rounding would make results vary from run to run on fits whose outcome
// hangs on the last bit.  Hence (a) every sum over the rows of a fit is taken in the order the cooperative kernel's
// DPP all-reduce imposes -- leaf l = term(row l) [+ term(row l + 16) for l < 5, fused], then the balanced tree
// ((l0 + l1) + (l2 + l3)) + ... over the 16 leaves -- and (b) every multiply-add whose contraction the compiler could
// decide differently in the two kernels is spelled out: this part of the file is compiled with contraction off, so
// only the fma()s written here (and in mpx_lm.hpp) fuse.
__device__ __forceinline__ double prod(double a, double b) { return a * b; }
__device__ __forceinline__ double fma_as_written(double a, double b, double c) { return __builtin_fma(a, b, c); }
// leaf l of a row sum: p = x_l y_l (0 for rows above `from`), plus x_{l+16} y_{l+16} where that row can exist
__device__ __forceinline__ double row_leaf(bool live, double xl, double yl) { return live ? prod(xl, yl) : 0.0; }
__device__ __forceinline__ double tree16(const double* t) {
    const double q0 = (t[0] + t[1]) + (t[2] + t[3]), q1 = (t[4] + t[5]) + (t[6] + t[7]);
    const double q2 = (t[8] + t[9]) + (t[10] + t[11]), q3 = (t[12] + t[13]) + (t[14] + t[15]);
    return (q0 + q1) + (q2 + q3);
}
// sum_{i >= from} x(i) y(i) over the lm::MAXM rows a lane of peakfit_kernel holds (rows >= m are exact zeros)
template <typename FX, typename FY>
__device__ __forceinline__ double dot_rows(int from, FX x, FY y) {
    static_assert(lm::MAXM > 16 && lm::MAXM <= 21, "rows 16.. pair with leaves 0..4");
    auto leaf = [&](int l) {
        const double p = row_leaf(l >= from, x(l), y(l));
        return l + 16 < lm::MAXM ? fma_as_written(x(l + 16), y(l + 16), p) : p;
    };
    // the balanced tree of tree16, quad by quad (four leaves live at a time, not sixteen)
    double q[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = (leaf(4 * k) + leaf(4 * k + 1)) + (leaf(4 * k + 2) + leaf(4 * k + 3));
    return (q[0] + q[1]) + (q[2] + q[3]);
}

struct GaussEval {
    double ampl, mu, ninv;  // ninv = -1 / (2 dev^2 + eps)   (peakutils.gaussian)
    const double* tab;      // 2^(j/64), LDS
};
__device__ __forceinline__ GaussEval gauss_prep(const double* p, const double* tab) {
    return {p[0], p[1], -lm::lm_rcp(fma_as_written(2.0, prod(p[2], p[2]), lm::EPSMCH)), tab};
}
__device__ __forceinline__ double gauss_resid(const GaussEval& g, double xi, double yi) {
    const double d = xi - g.mu;
    return fma_as_written(g.ampl, exp_nonpos(prod(d, d) * g.ninv, g.tab), -yi);
}
// ---- forward-difference columns of the centre and the width without their exponentials (round 5) -----------------------
// MINPACK's fdjac2 column j is (f(x + h e_j) - f(x)) / h.  For the gaussian f(x) + y = A e^u, u = d^2 ninv (d = x_i - centre), and
// the perturbed argument is u + t with
//   centre:  t = ((d - h)^2 - d^2) ninv = h (h - 2 d) ninv;       width:  t = d^2 (ninv' - ninv),  ninv' - ninv = 2 h (2 s + h) ninv ninv'
// (h the step as taken, fl(x_j + h) - x_j), so f(x + h e_j) - f(x) = (f + y)(e^t - 1) -- the amplitude column's own f + y, one
// exponential per row instead of three.  e^t - 1 by its series while |t| <= 2^-6 (relative error 1.3e-12: four orders below the
// rounding noise eps / h = 1e-8 of the difference it replaces); a fit with a row outside that range takes the two
// exponentials as before (fd_big, decided per FIT from its own scalars: the three fit kernels agree on it and stay bit-equal).
constexpr double FD_SERIES_MAX = 0.015625;
__device__ __forceinline__ double expm1_small(double t) {
    double p = fma_as_written(t, 1.0 / 120.0, 1.0 / 24.0);
    p = fma_as_written(t, p, 1.0 / 6.0);
    p = fma_as_written(t, p, 0.5);
    p = fma_as_written(t, p, 1.0);
    return prod(t, p);
}
struct FdStep {
    double h1, k1, inv_h1;   // centre: the step as taken, h1 ninv, 1 / the nominal step (what fdjac2 divides by)
    double dn, inv_h2;       // width: ninv' - ninv, 1 / the nominal step
    double mu1, ninv2;       // the perturbed centre and -1 / (2 s'^2 + eps) themselves (slow path)
};
__device__ __forceinline__ FdStep fd_prep(const double* x, double ninv, double eps) {
    FdStep s;
    double h1 = eps * fabs(x[1]), h2 = eps * fabs(x[2]);
    if (h1 == 0.0) h1 = eps;
    if (h2 == 0.0) h2 = eps;
    s.mu1 = x[1] + h1;
    s.h1 = s.mu1 - x[1];
    s.k1 = prod(s.h1, ninv);
    s.inv_h1 = lm::lm_rcp(h1);
    const double s2 = x[2] + h2, h2e = s2 - x[2];
    s.ninv2 = -lm::lm_rcp(fma_as_written(2.0, prod(s2, s2), lm::EPSMCH));
    s.dn = prod(prod(prod(2.0, h2e), x[2] + s2), prod(ninv, s.ninv2));
    s.inv_h2 = lm::lm_rcp(h2);
    return s;
}
// one row: t = f + y, d = x_i - centre
__device__ __forceinline__ void fd_row(const FdStep& s, double d, double t, double& j1, double& j2) {
    const double t1 = prod(fma_as_written(-2.0, d, s.h1), s.k1), t2 = prod(prod(d, d), s.dn);
    j1 = prod(prod(t, expm1_small(t1)), s.inv_h1);
    j2 = prod(prod(t, expm1_small(t2)), s.inv_h2);
}
// Does some row of the fit leave the series' range?  |t| is largest at the first or the last row (|linear| and a square are
// convex in d): decided from the fit's own scalars, the same in every kernel and in every lane of a cooperative fit.  (NaN: yes.)
__device__ __forceinline__ bool fd_big(const FdStep& s, double d_first, double d_last) {
    const double a = fmax(fabs(prod(fma_as_written(-2.0, d_first, s.h1), s.k1)), fabs(prod(fma_as_written(-2.0, d_last, s.h1), s.k1)));
    const double b = fmax(fabs(prod(prod(d_first, d_first), s.dn)), fabs(prod(prod(d_last, d_last), s.dn)));
    return !(a <= FD_SERIES_MAX && b <= FD_SERIES_MAX);
}

__device__ __forceinline__ void load_samples(const double* __restrict__ row, int m, double* ys) {
#pragma unroll
    for (int q = 0; q < lm::MAXM; ++q) ys[q] = q < m ? row[q] : 0.0;
}

// MINPACK state of a fit at the top of lmdif's outer loop (fvec is recomputed from x: same function, same bits).
struct ParkedFit {
    long long out;      // slot in center[] / ok[]
    long long row_off;  // first sample of the 21-sample window, relative to y
    double x0;          // abscissa of that sample
    double x[3], diag[3];
    double par, delta, xnorm, fnorm;
    int m, it, nfev, pad;
};
constexpr int PARK_NFEV_SMALL = 100;   // batches below 2048 frames (one clip: 2.57 instead of 2.80 ms): nothing to wait for, the cooperative trips are the faster ones
constexpr int PARK_NFEV = 160;   // (swept 60 ... 550 on three workloads: 130-180 is the flat optimum)
constexpr int PARK_LIVE = 8;   // park from waves with at most this many unfinished fits ...
constexpr int COOP_THREADS = 256, COOP_PAD_KB = 84;   // the cooperative kernels: one workgroup of four waves per CU (a wave per SIMD), held apart by unused LDS (more than half of the 160 KB; what is left takes a 64 KB workgroup of another context)
constexpr int COOP_SPLIT = 4096;        // parked lists longer than this -- more than a wave per SIMD at four fits each -- run eight fits to a wave (coopfit8_kernel)
constexpr int COOP_PASS1_TRIPS = 24;   // coopfit_kernel, first pass: trips after which a fit still open is parked again
constexpr int EARLY_PARK_SLOW_NFEV = 100;   // ... and the slow ones (2 % of the fits need 100 ... 400 evaluations; the median is 56)
constexpr int EARLY_PARK_CAP = 12288;   // round 6, development builds: budget of the early hand-off (fits that look like runaways leave the lane kernel at once; peakfit_kernel)
constexpr int PARK_CAP = 16384;  // ... or while fewer fits than this have asked (about what coopfit_kernel holds at once)

// Cross-lane traffic of the cooperative fit on DPP (register-to-register, ~8 cycles) instead of ds_bpermute
// (~100 cycles through the LDS crossbar): a fit owns one 16-lane DPP row, so its all-reduce is four mirror /
// quad-permute steps.  Every step pairs lanes symmetrically (i <-> partner(i)), so both partners add the same
// two numbers and all 16 lanes end with identical bits.
// (dpp_f64: above block_minmax)
__device__ __forceinline__ double row_sum(double v) {
    v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);   // row_half_mirror
    v += dpp_f64<0x140>(v);   // row_mirror
    return v;
}
// value of lane J of the caller's 16-lane row: DPP row_newbcast, register to register.  (Until round 3: ds_swizzle in
// bit-mask mode, a round trip through the LDS crossbar with a wait, thirty of them on the dependent path of a trip.)
template <int J>
__device__ __forceinline__ double row_bcast_c(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + J, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + J, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_bcast(double v, int j) {  // j in 0..2, a constant after unrolling
    return j == 0 ? row_bcast_c<0>(v) : (j == 1 ? row_bcast_c<1>(v) : row_bcast_c<2>(v));
}
struct D2 {  // this lane's two samples: rows l and l + 16 of the m <= 21 rows
    double a, b;
};
// this lane's leaf of sum_{i >= from} x(i) y(i) (see dot_rows): row l counts from `from` on, row l + 16 exists for l < 5
__device__ __forceinline__ double coop_leaf(int l, int from, D2 x, D2 y) {
    const double p = row_leaf(l >= from, x.a, y.a);
    const double f = fma_as_written(x.b, y.b, p);
    return l + 16 < lm::MAXM ? f : p;
}

// Cooperative continuation of parked fits: one fit per 16-lane row (4 per wave), lane l holding samples l and
// l + 16 (lm::gaussian_fit restated from a resume point, m-vectors across lanes, with the same reciprocal / column-0 / exponential
// forms as peakfit_kernel).  The m-vectors of MINPACK are two registers per lane, norms and dot products are
// all-reduces that leave identical bits in every lane of the row, so the row runs the 3x3 part redundantly and
// uniformly.  A trip costs ~3.3 k instructions per FOUR fits instead of ~8.5 k and, more to the point, every
// parked fit gets its own lanes instead of waiting behind 63 finished ones.
// wave sum of a per-lane count into a 64-bit counter kept as two 32-bit words (one atomic per wave)
__device__ __forceinline__ void count_evals(int* counter, unsigned mine) {
    unsigned long long v = mine;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0 && v) {
        const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
        const unsigned old = atomicAdd(reinterpret_cast<unsigned*>(counter), lo);
        const unsigned carry = (old + lo < old) ? 1u : 0u;
        if (hi + carry) atomicAdd(reinterpret_cast<unsigned*>(counter) + 1, hi + carry);
    }
}

// Two passes (round 4).  A batch parks thousands of fits; many need a few more trips, the rest burn MINPACK's whole budget
// (160 more trips).  Pass 1 (`repark` set) stops every fit still open after `max_trips` trips at the top of lmdif's outer loop and
// parks it again (the same 128-byte state); pass 2 takes the survivors, again four to a wave: the short fits are gone, the long
// ones sit densely packed in fewer waves, and fewer waves share a SIMD's fp64 issue (clip batches: 1.9 -> 1.65 and 3.2 -> 2.55
// ms; a batch whose parked fits all survive pays the second launch, 0.05 ms).  Same arithmetic, same bits.  `spread` (measured,
// not used: development knob) gives the survivors a wave each first -- fit i to row i / waves of wave i % waves: SLOWER, 1.06 ->
// 1.70 ms for 512 fits; a wave of one live row costs what a wave of four does, and four times as many waves are busy.
__global__ __launch_bounds__(256) void coopfit_kernel(const ParkedFit* __restrict__ parked, const int* __restrict__ parked_count,
                                                      int* next_parked, const double* __restrict__ y, double* center,
                                                      int* ok, int maxfev, ParkedFit* repark, int* repark_count, int max_trips,
                                                      int spread, int* evals, int min_total, int max_total) {
    using namespace lm;
    const int total = *parked_count;
    if (total < min_total || total > max_total) return;   // the other form of the kernel takes this list (see esacf_run)
    __shared__ double exp_tab[64];
    if (threadIdx.x < 64) exp_tab[threadIdx.x] = exp2((double)threadIdx.x * (1.0 / 64.0));
    __syncthreads();
    const int l = threadIdx.x & 15;
    unsigned my_evals = 0;
    const double ftol = 1.49012e-8, xtol = 1.49012e-8, gtol = 0.0, factor = 100.0;
    const double eps = sqrt(EPSMCH);
    bool first = spread != 0;
    for (;;) {
        int idx = 0;
        if (first) {
            idx = (int)(threadIdx.x >> 4) * (int)gridDim.x + (int)blockIdx.x;
            first = false;
        } else {
            if (l == 0) idx = atomicAdd(next_parked, 1);
            idx = __shfl(idx, 0, 16) + (spread ? (int)(blockDim.x >> 4) * (int)gridDim.x : 0);
        }
        if (idx >= total) break;
        const ParkedFit pf = parked[idx];
        const bool ona = l < pf.m, onb = l + 16 < pf.m;
        const D2 px = {pf.x0 + (double)l, pf.x0 + (double)(l + 16)};
        const D2 py = {ona ? y[pf.row_off + l] : 0.0, onb ? y[pf.row_off + l + 16] : 0.0};
        double x[NP] = {pf.x[0], pf.x[1], pf.x[2]}, diag[NP] = {pf.diag[0], pf.diag[1], pf.diag[2]};
        double par = pf.par, delta = pf.delta, xnorm = pf.xnorm, fnorm = pf.fnorm;
        int it = pf.it, nfev = pf.nfev, info = 0;
        auto resid = [&](const double* p) -> D2 {
            const GaussEval g = gauss_prep(p, exp_tab);
            const double ra = gauss_resid(g, px.a, py.a), rb = gauss_resid(g, px.b, py.b);
            return {ona ? ra : 0.0, onb ? rb : 0.0};
        };
        D2 f = resid(x);
        int trips = 0;
        bool again = false;
        // A trip = at most one OUTER step (jacobian, QR: only the fits whose last trial was accepted) and ONE trial (round 5).
        // Until then a trip ran the trial loop to acceptance for every fit of the wave: one fit's rejected trial (21 % of the
        // trials) made the other fits wait for a second `lmpar` -- 9 k of a trip's 26 k clocks on 85 % of the trips at eight fits
        // to a wave.  The same operations per fit in the same order: the same bits.  What a trial needs of the OUTER step (R, Q^T f,
        // the pivots, gnorm) stays in registers across trips.
        bool need_outer = true;
        int ipvt[NP] = {0, 1, 2};
        double qtf[NP] = {0.0, 0.0, 0.0}, r[NP * NP], gnorm = 0.0;
#pragma unroll
        for (int i = 0; i < NP * NP; ++i) r[i] = 0.0;
        for (;;) {
            if (need_outer) {
            if (repark != nullptr && trips >= max_trips) {   // (uniform in the row; a fit is parked again at the top of lmdif's outer loop only)
                again = true;
                break;
            }
            ++trips;
            // forward-difference jacobian: this lane's two rows
            D2 J0, J1, J2;
            {
                if (x[0] != 0.0) {
                    const double inv_a = lm_rcp(x[0]);
                    J0 = {ona ? (f.a + py.a) * inv_a : 0.0, onb ? (f.b + py.b) * inv_a : 0.0};
                } else {
                    x[0] = eps;
                    const D2 w = resid(x);
                    J0 = {(w.a - f.a) * (1.0 / eps), (w.b - f.b) * (1.0 / eps)};
                    x[0] = 0.0;
                }
                const FdStep fs = fd_prep(x, gauss_prep(x, exp_tab).ninv, eps);
                if (!fd_big(fs, pf.x0 - x[1], pf.x0 + (double)(pf.m - 1) - x[1])) {
                    fd_row(fs, px.a - x[1], f.a + py.a, J1.a, J2.a);
                    fd_row(fs, px.b - x[1], f.b + py.b, J1.b, J2.b);
                    J1 = {ona ? J1.a : 0.0, onb ? J1.b : 0.0};
                    J2 = {ona ? J2.a : 0.0, onb ? J2.b : 0.0};
                } else {   // a row outside the series' range (a width far below a lag): the two exponentials per row
                    const double keep1 = x[1], keep2 = x[2];
                    x[1] = fs.mu1;
                    const D2 w1 = resid(x);
                    x[1] = keep1;
                    x[2] = keep2 + (eps * fabs(keep2) == 0.0 ? eps : eps * fabs(keep2));
                    const D2 w2 = resid(x);
                    x[2] = keep2;
                    J1 = {(w1.a - f.a) * fs.inv_h1, (w1.b - f.b) * fs.inv_h1};
                    J2 = {(w2.a - f.a) * fs.inv_h2, (w2.b - f.b) * fs.inv_h2};
                }
            }
            nfev += NP;
            ipvt[0] = 0;
            ipvt[1] = 1;
            ipvt[2] = 2;
            double acnorm[NP], rdiag[NP], wa[NP];
            acnorm[0] = lm_sqrt(row_sum(coop_leaf(l, 0, J0, J0)));
            acnorm[1] = lm_sqrt(row_sum(coop_leaf(l, 0, J1, J1)));
            acnorm[2] = lm_sqrt(row_sum(coop_leaf(l, 0, J2, J2)));
#pragma unroll
            for (int j = 0; j < NP; ++j) rdiag[j] = wa[j] = acnorm[j];
            D2 w4 = f;  // becomes Q^T fvec
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                int kmax = j;
                double rmax = rdiag[j];
#pragma unroll
                for (int k = j + 1; k < NP; ++k)
                    if (rdiag[k] > rmax) {
                        kmax = k;
                        rmax = rdiag[k];
                    }
                if (kmax != j) {
                    D2& cjs = j == 0 ? J0 : (j == 1 ? J1 : J2);
                    D2& cks = kmax == 1 ? J1 : J2;
                    const D2 t0 = cjs;
                    cjs = cks;
                    cks = t0;
                    put3(rdiag, kmax, rdiag[j]);
                    put3(wa, kmax, wa[j]);
                    const int t = ipvt[j];
                    ipvt[j] = sel3(ipvt, kmax);
                    put3(ipvt, kmax, t);
                }
                D2& cj = j == 0 ? J0 : (j == 1 ? J1 : J2);
                const bool below = l >= j;  // rows j..15 of the first slot; the second slot (rows 16..) is always below
                double ajnorm = lm_sqrt(row_sum(coop_leaf(l, j, cj, cj)));
                if (ajnorm != 0.0) {
                    if (row_bcast(cj.a, j) < 0.0) ajnorm = -ajnorm;
                    const double inv_aj = lm_rcp(ajnorm);
                    if (below) cj.a *= inv_aj;
                    cj.b *= inv_aj;
                    if (l == j) cj.a += 1.0;
                    const double inv_ajj = lm_rcp(row_bcast(cj.a, j));
#pragma unroll
                    for (int k = j + 1; k < NP; ++k) {
                        D2& ck = k == 1 ? J1 : J2;
                        const double temp = row_sum(coop_leaf(l, j, cj, ck)) * inv_ajj;
                        if (below) ck.a = fma_as_written(-temp, cj.a, ck.a);
                        ck.b = fma_as_written(-temp, cj.b, ck.b);
                        if (rdiag[k] != 0.0) {
                            const double t = lm_div(row_bcast(ck.a, j), rdiag[k]);
                            const double u = fma(-t, t, 1.0);
                            rdiag[k] *= lm_sqrt(u > 0.0 ? u : 0.0);
                            const double q = lm_div(rdiag[k], wa[k]);
                            if (0.05 * q * q <= EPSMCH) {
                                rdiag[k] = lm_sqrt(row_sum(coop_leaf(l, j + 1, ck, ck)));
                                wa[k] = rdiag[k];
                            }
                        }
                    }
                    const double temp = -row_sum(coop_leaf(l, j, cj, w4)) * inv_ajj;
                    if (below) w4.a = fma_as_written(cj.a, temp, w4.a);
                    w4.b = fma_as_written(cj.b, temp, w4.b);
                }
                rdiag[j] = -ajnorm;
                qtf[j] = row_bcast(w4.a, j);
            }
            if (it == 1) {
                double wa3[NP];
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    diag[j] = acnorm[j] != 0.0 ? acnorm[j] : 1.0;
                    wa3[j] = diag[j] * x[j];
                }
                xnorm = enorm3(wa3);
                delta = factor * xnorm;
                if (delta == 0.0) delta = factor;
            }
            // replicate the 3x3 upper triangle R (row i lives in lane i, first slot; its diagonal is rdiag)
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                r[i * NP + 0] = i == 0 ? rdiag[0] : row_bcast(J0.a, i);
                r[i * NP + 1] = i == 1 ? rdiag[1] : row_bcast(J1.a, i);
                r[i * NP + 2] = i == 2 ? rdiag[2] : row_bcast(J2.a, i);
            }
            gnorm = 0.0;
            if (fnorm != 0.0) {
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const double an = sel3(acnorm, ipvt[j]);
                    if (an != 0.0) {
                        double s2 = 0.0;
#pragma unroll
                        for (int i = 0; i <= j; ++i) s2 = fma(r[i * NP + j], lm_div(qtf[i], fnorm), s2);
                        const double g = fabs(lm_div(s2, an));
                        gnorm = g > gnorm ? g : gnorm;
                    }
                }
            }
            if (gnorm <= gtol) {
                info = 4;
                break;
            }
#pragma unroll
            for (int j = 0; j < NP; ++j) diag[j] = diag[j] > acnorm[j] ? diag[j] : acnorm[j];
            }
            {
                double rr[NP * NP], p[NP], xnew[NP], wa3[NP], sd[NP];
#pragma unroll
                for (int i = 0; i < NP * NP; ++i) rr[i] = r[i];
                par = lmpar(rr, ipvt, diag, qtf, delta, par, p, sd);
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    p[j] = -p[j];
                    xnew[j] = x[j] + p[j];
                    wa3[j] = diag[j] * p[j];
                }
                const double pnorm = enorm3(wa3);
                if (it == 1) delta = delta < pnorm ? delta : pnorm;
                const D2 fn = resid(xnew);
                ++nfev;
                const double fnorm1 = lm_sqrt(row_sum(coop_leaf(l, 0, fn, fn)));
                double actred = -1.0;
                if (0.1 * fnorm1 < fnorm) {
                    const double q = lm_div(fnorm1, fnorm);
                    actred = fma(-q, q, 1.0);
                }
#pragma unroll
                for (int j = 0; j < NP; ++j) wa3[j] = 0.0;
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const double temp = sel3(p, ipvt[j]);
#pragma unroll
                    for (int i = 0; i <= j; ++i) wa3[i] = fma(r[i * NP + j], temp, wa3[i]);
                }
                const double temp1 = lm_div(enorm3(wa3), fnorm);
                const double temp2 = lm_div(lm_sqrt(par) * pnorm, fnorm);
                const double prered = fma(temp1, temp1, temp2 * temp2 / 0.5);
                const double dirder = -fma(temp1, temp1, temp2 * temp2);
                const double ratio = prered != 0.0 ? lm_div(actred, prered) : 0.0;
                if (ratio <= 0.25) {
                    double temp = actred >= 0.0 ? 0.5 : lm_div(0.5 * dirder, dirder + 0.5 * actred);
                    if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
                    const double p10 = lm_div(pnorm, 0.1), dm = delta < p10 ? delta : p10;
                    delta = temp * dm;
                    par = lm_div(par, temp);
                } else if (par == 0.0 || ratio >= 0.75) {
                    delta = pnorm / 0.5;
                    par = 0.5 * par;
                }
                if (ratio >= 1e-4) {
#pragma unroll
                    for (int j = 0; j < NP; ++j) {
                        x[j] = xnew[j];
                        wa3[j] = diag[j] * x[j];
                    }
                    f = fn;
                    xnorm = enorm3(wa3);
                    fnorm = fnorm1;
                    ++it;
                }
                const bool c1 = fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0;
                if (c1) info = 1;
                if (delta <= xtol * xnorm) info = 2;
                if (c1 && info == 2) info = 3;
                if (info != 0) break;
                if (nfev >= maxfev) info = 5;
                if (fabs(actred) <= EPSMCH && prered <= EPSMCH && 0.5 * ratio <= 1.0) info = 6;
                if (delta <= EPSMCH * xnorm) info = 7;
                if (gnorm <= EPSMCH) info = 8;
                if (info != 0) break;
                need_outer = ratio >= 1e-4;   // accepted: a new jacobian next trip; rejected: the next trial with the same R
            }
        }
        if (l == 0) {
            if (again) {
                ParkedFit* q = repark + atomicAdd(repark_count, 1);   // (field by field: a local copy of the record cost a stack slot)
                q->out = pf.out;
                q->row_off = pf.row_off;
                q->x0 = pf.x0;
                q->x[0] = x[0];
                q->x[1] = x[1];
                q->x[2] = x[2];
                q->diag[0] = diag[0];
                q->diag[1] = diag[1];
                q->diag[2] = diag[2];
                q->par = par;
                q->delta = delta;
                q->xnorm = xnorm;
                q->fnorm = fnorm;
                q->m = pf.m;
                q->it = it;
                q->nfev = nfev;
                q->pad = 0;
            } else {
                ok[pf.out] = (info >= 1 && info <= 4) ? 1 : 0;
                center[pf.out] = x[1];
            }
            my_evals += (unsigned)(nfev - pf.nfev);
        }
    }
    count_evals(evals, my_evals);   // total[6..7]: function evaluations of the batch (statistics only)
}

// ---- the same fit on EIGHT lanes (round 4): coopfit8_kernel ---------------------------------------------------------------
// The cooperative trips are bound by the SIMD's fp64 issue (a lone wave runs its 3.3 k instructions per trip at 4.8 clocks
// each; two waves on a SIMD take twice as long per trip), and the 3x3 algebra of a trip is replicated in every lane whatever
// the number of fits in the wave: eight fits per wave -- lane l of a fit's eight holds rows l, l + 8 and l + 16 -- are half
// the busy waves for 1.3x the instructions.  The sums keep the order of dot_rows / coop_leaf bit for bit: slot A (+ C, fused)
// is leaf l, slot B leaf l + 8 of the 16-leaf tree; quad_perm, quad_perm and row_half_mirror reduce the leaves 0..7 to
// q0 + q1 and the leaves 8..15 to q2 + q3 in every lane of the eight, and their sum is (q0 + q1) + (q2 + q3).
struct D3 {  // this lane's three samples: rows l, l + 8 and l + 16 of the m <= 21 rows
    double a, b, c;
};
__device__ __forceinline__ double half_sum(double v) {
    v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);   // row_half_mirror
    return v;
}
__device__ __forceinline__ double coop8_sum(int l, int from, D3 x, D3 y) {
    const double p = row_leaf(l >= from, x.a, y.a);
    const double f = fma_as_written(x.c, y.c, p);
    const double leaf_lo = l + 16 < lm::MAXM ? f : p;   // leaf l: rows l and l + 16
    const double leaf_hi = prod(x.b, y.b);              // leaf l + 8: row l + 8 (never above `from` <= 3)
    return half_sum(leaf_lo) + half_sum(leaf_hi);
}
// value of lane J of the caller's eight: broadcast inside each quad, then the quads 1 and 3 of the DPP row take the
// value of the quad to their left (row_shr:4 written to banks 1 and 3 only)
template <int J>
__device__ __forceinline__ double half_bcast_c(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, J * 0x55, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, J * 0x55, 0xf, 0xf, false);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x114, 0xf, 0xa, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x114, 0xf, 0xa, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double half_bcast(double v, int j) {  // j in 0..2, a constant after unrolling
    return j == 0 ? half_bcast_c<0>(v) : (j == 1 ? half_bcast_c<1>(v) : half_bcast_c<2>(v));
}

__global__ __launch_bounds__(256) void coopfit8_kernel(const ParkedFit* __restrict__ parked, const int* __restrict__ parked_count,
                                                     int* next_parked, const double* __restrict__ y, double* center,
                                                     int* ok, int maxfev, ParkedFit* repark, int* repark_count, int max_trips,
                                                     int spread, int* evals, int min_total, int max_total) {
    using namespace lm;
    const int total = *parked_count;
    if (total < min_total || total > max_total) return;
    __shared__ double exp_tab[64];
    if (threadIdx.x < 64) exp_tab[threadIdx.x] = exp2((double)threadIdx.x * (1.0 / 64.0));
    __syncthreads();
    const int l = threadIdx.x & 7;
    unsigned my_evals = 0;
    const double ftol = 1.49012e-8, xtol = 1.49012e-8, gtol = 0.0, factor = 100.0;
    const double eps = sqrt(EPSMCH);
    bool first = spread != 0;
    for (;;) {
        int idx = 0;
        if (first) {
            idx = (int)(threadIdx.x >> 3) * (int)gridDim.x + (int)blockIdx.x;   // (development knob, see coopfit_kernel)
            first = false;
        } else {
            if (l == 0) idx = atomicAdd(next_parked, 1);
            idx = __shfl(idx, 0, 8) + (spread ? (int)(blockDim.x >> 3) * (int)gridDim.x : 0);
        }
        if (idx >= total) break;
        const ParkedFit pf = parked[idx];
        const bool ona = l < pf.m, onb = l + 8 < pf.m, onc = l + 16 < pf.m;
        const D3 px = {pf.x0 + (double)l, pf.x0 + (double)(l + 8), pf.x0 + (double)(l + 16)};
        const D3 py = {ona ? y[pf.row_off + l] : 0.0, onb ? y[pf.row_off + l + 8] : 0.0, onc ? y[pf.row_off + l + 16] : 0.0};
        double x[NP] = {pf.x[0], pf.x[1], pf.x[2]}, diag[NP] = {pf.diag[0], pf.diag[1], pf.diag[2]};
        double par = pf.par, delta = pf.delta, xnorm = pf.xnorm, fnorm = pf.fnorm;
        int it = pf.it, nfev = pf.nfev, info = 0;
        auto resid = [&](const double* p) -> D3 {
            const GaussEval g = gauss_prep(p, exp_tab);
            const double ra = gauss_resid(g, px.a, py.a), rb = gauss_resid(g, px.b, py.b), rc = gauss_resid(g, px.c, py.c);
            return {ona ? ra : 0.0, onb ? rb : 0.0, onc ? rc : 0.0};
        };
        D3 f = resid(x);
        int trips = 0;
        bool again = false;
        // A trip = at most one OUTER step (jacobian, QR: only the fits whose last trial was accepted) and ONE trial (round 5).
        // Until then a trip ran the trial loop to acceptance for every fit of the wave: one fit's rejected trial (21 % of the
        // trials) made the other fits wait for a second `lmpar` -- 9 k of a trip's 26 k clocks on 85 % of the trips at eight fits
        // to a wave.  The same operations per fit in the same order: the same bits.  What a trial needs of the OUTER step (R, Q^T f,
        // the pivots, gnorm) stays in registers across trips.
        bool need_outer = true;
        int ipvt[NP] = {0, 1, 2};
        double qtf[NP] = {0.0, 0.0, 0.0}, r[NP * NP], gnorm = 0.0;
#pragma unroll
        for (int i = 0; i < NP * NP; ++i) r[i] = 0.0;
        for (;;) {
            if (need_outer) {
            if (repark != nullptr && trips >= max_trips) {   // (uniform in the row; a fit is parked again at the top of lmdif's outer loop only)
                again = true;
                break;
            }
            ++trips;
            // forward-difference jacobian: this lane's three rows
            D3 J0, J1, J2;
            {
                if (x[0] != 0.0) {
                    const double inv_a = lm_rcp(x[0]);
                    J0 = {ona ? (f.a + py.a) * inv_a : 0.0, onb ? (f.b + py.b) * inv_a : 0.0, onc ? (f.c + py.c) * inv_a : 0.0};
                } else {
                    x[0] = eps;
                    const D3 w = resid(x);
                    J0 = {(w.a - f.a) * (1.0 / eps), (w.b - f.b) * (1.0 / eps), (w.c - f.c) * (1.0 / eps)};
                    x[0] = 0.0;
                }
                const FdStep fs = fd_prep(x, gauss_prep(x, exp_tab).ninv, eps);
                if (!fd_big(fs, pf.x0 - x[1], pf.x0 + (double)(pf.m - 1) - x[1])) {
                    fd_row(fs, px.a - x[1], f.a + py.a, J1.a, J2.a);
                    fd_row(fs, px.b - x[1], f.b + py.b, J1.b, J2.b);
                    fd_row(fs, px.c - x[1], f.c + py.c, J1.c, J2.c);
                    J1 = {ona ? J1.a : 0.0, onb ? J1.b : 0.0, onc ? J1.c : 0.0};
                    J2 = {ona ? J2.a : 0.0, onb ? J2.b : 0.0, onc ? J2.c : 0.0};
                } else {   // a row outside the series' range (a width far below a lag): the two exponentials per row
                    const double keep1 = x[1], keep2 = x[2];
                    x[1] = fs.mu1;
                    const D3 w1 = resid(x);
                    x[1] = keep1;
                    x[2] = keep2 + (eps * fabs(keep2) == 0.0 ? eps : eps * fabs(keep2));
                    const D3 w2 = resid(x);
                    x[2] = keep2;
                    J1 = {(w1.a - f.a) * fs.inv_h1, (w1.b - f.b) * fs.inv_h1, (w1.c - f.c) * fs.inv_h1};
                    J2 = {(w2.a - f.a) * fs.inv_h2, (w2.b - f.b) * fs.inv_h2, (w2.c - f.c) * fs.inv_h2};
                }
            }
            nfev += NP;
            ipvt[0] = 0;
            ipvt[1] = 1;
            ipvt[2] = 2;
            double acnorm[NP], rdiag[NP], wa[NP];
            acnorm[0] = lm_sqrt(coop8_sum(l, 0, J0, J0));
            acnorm[1] = lm_sqrt(coop8_sum(l, 0, J1, J1));
            acnorm[2] = lm_sqrt(coop8_sum(l, 0, J2, J2));
#pragma unroll
            for (int j = 0; j < NP; ++j) rdiag[j] = wa[j] = acnorm[j];
            D3 w4 = f;  // becomes Q^T fvec
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                int kmax = j;
                double rmax = rdiag[j];
#pragma unroll
                for (int k = j + 1; k < NP; ++k)
                    if (rdiag[k] > rmax) {
                        kmax = k;
                        rmax = rdiag[k];
                    }
                if (kmax != j) {
                    D3& cjs = j == 0 ? J0 : (j == 1 ? J1 : J2);
                    D3& cks = kmax == 1 ? J1 : J2;
                    const D3 t0 = cjs;
                    cjs = cks;
                    cks = t0;
                    put3(rdiag, kmax, rdiag[j]);
                    put3(wa, kmax, wa[j]);
                    const int t = ipvt[j];
                    ipvt[j] = sel3(ipvt, kmax);
                    put3(ipvt, kmax, t);
                }
                D3& cj = j == 0 ? J0 : (j == 1 ? J1 : J2);
                const bool below = l >= j;  // rows j..7 of the first slot; the other two slots (rows 8.., 16..) are always below
                double ajnorm = lm_sqrt(coop8_sum(l, j, cj, cj));
                if (ajnorm != 0.0) {
                    if (half_bcast(cj.a, j) < 0.0) ajnorm = -ajnorm;
                    const double inv_aj = lm_rcp(ajnorm);
                    if (below) cj.a *= inv_aj;
                    cj.b *= inv_aj;
                    cj.c *= inv_aj;
                    if (l == j) cj.a += 1.0;
                    const double inv_ajj = lm_rcp(half_bcast(cj.a, j));
#pragma unroll
                    for (int k = j + 1; k < NP; ++k) {
                        D3& ck = k == 1 ? J1 : J2;
                        const double temp = coop8_sum(l, j, cj, ck) * inv_ajj;
                        if (below) ck.a = fma_as_written(-temp, cj.a, ck.a);
                        ck.b = fma_as_written(-temp, cj.b, ck.b);
                        ck.c = fma_as_written(-temp, cj.c, ck.c);
                        if (rdiag[k] != 0.0) {
                            const double t = lm_div(half_bcast(ck.a, j), rdiag[k]);
                            const double u = fma(-t, t, 1.0);
                            rdiag[k] *= lm_sqrt(u > 0.0 ? u : 0.0);
                            const double q = lm_div(rdiag[k], wa[k]);
                            if (0.05 * q * q <= EPSMCH) {
                                rdiag[k] = lm_sqrt(coop8_sum(l, j + 1, ck, ck));
                                wa[k] = rdiag[k];
                            }
                        }
                    }
                    const double temp = -coop8_sum(l, j, cj, w4) * inv_ajj;
                    if (below) w4.a = fma_as_written(cj.a, temp, w4.a);
                    w4.b = fma_as_written(cj.b, temp, w4.b);
                    w4.c = fma_as_written(cj.c, temp, w4.c);
                }
                rdiag[j] = -ajnorm;
                qtf[j] = half_bcast(w4.a, j);
            }
            if (it == 1) {
                double wa3[NP];
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    diag[j] = acnorm[j] != 0.0 ? acnorm[j] : 1.0;
                    wa3[j] = diag[j] * x[j];
                }
                xnorm = enorm3(wa3);
                delta = factor * xnorm;
                if (delta == 0.0) delta = factor;
            }
            // replicate the 3x3 upper triangle R (row i lives in lane i of the fit's eight, first slot; its diagonal is rdiag)
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                r[i * NP + 0] = i == 0 ? rdiag[0] : half_bcast(J0.a, i);
                r[i * NP + 1] = i == 1 ? rdiag[1] : half_bcast(J1.a, i);
                r[i * NP + 2] = i == 2 ? rdiag[2] : half_bcast(J2.a, i);
            }
            gnorm = 0.0;
            if (fnorm != 0.0) {
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const double an = sel3(acnorm, ipvt[j]);
                    if (an != 0.0) {
                        double s2 = 0.0;
#pragma unroll
                        for (int i = 0; i <= j; ++i) s2 = fma(r[i * NP + j], lm_div(qtf[i], fnorm), s2);
                        const double g = fabs(lm_div(s2, an));
                        gnorm = g > gnorm ? g : gnorm;
                    }
                }
            }
            if (gnorm <= gtol) {
                info = 4;
                break;
            }
#pragma unroll
            for (int j = 0; j < NP; ++j) diag[j] = diag[j] > acnorm[j] ? diag[j] : acnorm[j];
            }
            {
                double rr[NP * NP], p[NP], xnew[NP], wa3[NP], sd[NP];
#pragma unroll
                for (int i = 0; i < NP * NP; ++i) rr[i] = r[i];
#if defined(MPX_DEV_KNOBS) && defined(MPX_FIT_STATS)
                int lm_it = 0;
                par = lmpar(rr, ipvt, diag, qtf, delta, par, p, sd, &lm_it);
                if (l == 0) {
                    FIT_STAT(20 + lm_it, 1);
                    FIT_STAT(35, 1);
                }
                {
                    int wmax = 0;
                    for (int k = 1; k <= 10; ++k)
                        if (__ballot(lm_it >= k)) wmax = k;
                    const unsigned long long act = __ballot(1);
                    if ((int)(threadIdx.x & 63) == __ffsll((long long)act) - 1) {
                        FIT_STAT(32, 1);
                        FIT_STAT(33, wmax);
                        FIT_STAT(34, __popcll(act) >> 3);
                    }
                }
#else
                par = lmpar(rr, ipvt, diag, qtf, delta, par, p, sd);
#endif
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    p[j] = -p[j];
                    xnew[j] = x[j] + p[j];
                    wa3[j] = diag[j] * p[j];
                }
                const double pnorm = enorm3(wa3);
                if (it == 1) delta = delta < pnorm ? delta : pnorm;
                const D3 fn = resid(xnew);
                ++nfev;
                const double fnorm1 = lm_sqrt(coop8_sum(l, 0, fn, fn));
                double actred = -1.0;
                if (0.1 * fnorm1 < fnorm) {
                    const double q = lm_div(fnorm1, fnorm);
                    actred = fma(-q, q, 1.0);
                }
#pragma unroll
                for (int j = 0; j < NP; ++j) wa3[j] = 0.0;
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const double temp = sel3(p, ipvt[j]);
#pragma unroll
                    for (int i = 0; i <= j; ++i) wa3[i] = fma(r[i * NP + j], temp, wa3[i]);
                }
                const double temp1 = lm_div(enorm3(wa3), fnorm);
                const double temp2 = lm_div(lm_sqrt(par) * pnorm, fnorm);
                const double prered = fma(temp1, temp1, temp2 * temp2 / 0.5);
                const double dirder = -fma(temp1, temp1, temp2 * temp2);
                const double ratio = prered != 0.0 ? lm_div(actred, prered) : 0.0;
                if (ratio <= 0.25) {
                    double temp = actred >= 0.0 ? 0.5 : lm_div(0.5 * dirder, dirder + 0.5 * actred);
                    if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
                    const double p10 = lm_div(pnorm, 0.1), dm = delta < p10 ? delta : p10;
                    delta = temp * dm;
                    par = lm_div(par, temp);
                } else if (par == 0.0 || ratio >= 0.75) {
                    delta = pnorm / 0.5;
                    par = 0.5 * par;
                }
                if (ratio >= 1e-4) {
#pragma unroll
                    for (int j = 0; j < NP; ++j) {
                        x[j] = xnew[j];
                        wa3[j] = diag[j] * x[j];
                    }
                    f = fn;
                    xnorm = enorm3(wa3);
                    fnorm = fnorm1;
                    ++it;
                }
                const bool c1 = fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0;
                if (c1) info = 1;
                if (delta <= xtol * xnorm) info = 2;
                if (c1 && info == 2) info = 3;
                if (info != 0) break;
                if (nfev >= maxfev) info = 5;
                if (fabs(actred) <= EPSMCH && prered <= EPSMCH && 0.5 * ratio <= 1.0) info = 6;
                if (delta <= EPSMCH * xnorm) info = 7;
                if (gnorm <= EPSMCH) info = 8;
                if (info != 0) break;
                need_outer = ratio >= 1e-4;   // accepted: a new jacobian next trip; rejected: the next trial with the same R
            }
        }
        if (l == 0) {
            if (again) {
                ParkedFit* q = repark + atomicAdd(repark_count, 1);   // (field by field: a local copy of the record cost a stack slot)
                q->out = pf.out;
                q->row_off = pf.row_off;
                q->x0 = pf.x0;
                q->x[0] = x[0];
                q->x[1] = x[1];
                q->x[2] = x[2];
                q->diag[0] = diag[0];
                q->diag[1] = diag[1];
                q->diag[2] = diag[2];
                q->par = par;
                q->delta = delta;
                q->xnorm = xnorm;
                q->fnorm = fnorm;
                q->m = pf.m;
                q->it = it;
                q->nfev = nfev;
                q->pad = 0;
            } else {
                ok[pf.out] = (info >= 1 && info <= 4) ? 1 : 0;
                center[pf.out] = x[1];
            }
            my_evals += (unsigned)(nfev - pf.nfev);
        }
    }
    count_evals(evals, my_evals);   // total[6..7]: function evaluations of the batch (statistics only)
}



// ---- the cooperative fit as a LIVE consumer (round 6): coopfit_live_kernel -- MEASURED, NOT ADOPTED ----------------------
// Verdict (MI355X, the 8192-frame Target, profiles/r6/fit_live_ab*.txt; every variant bit-identical to the lane kernel alone,
// tests/tools/esacf_bitcheck.py: 0 differing rows): next to a lane kernel on HALF its grid (what leaves the registers for a
// wave of this kernel on every SIMD) with fits that look like runaways handed over from evaluation 20 on, the fits take 3.2-3.4
// ms against 2.85 for lane kernel + cooperative launches one after the other: a lone lane wave per SIMD is 1.6 x slower at
// the lane kernel's work, fits that start late in it and run away push the end out, and a cooperative trip slows down next to
// a lane wave.  Behind a FULL lane grid (this kernel's workgroups become resident as the lane kernel's exit) it takes the tail
// 0-3 % faster at 8192 frames (4.07-4.18 against 4.19-4.21 ms per call) and 13 % slower at 1024 -- not a result.  The lane
// kernel needs the whole register file for its throughput and the runaway fits' chain cannot start without registers of its
// own: the two phases stay one after the other.  What follows is the design as built.
// The runaway fits are ONE serial chain of ~200 LM steps each; the lane kernel is what the other 96 % of the fits need.  Until
// round 6 the chain waited for the lane kernel: a fit parked at evaluation 160 (0.8 ms into the kernel) was picked up when the
// kernel ended (1.3 ms).  This kernel runs NEXT TO the lane kernel, on a stream of its own, one wave per SIMD (the lane kernel
// keeps the other wave slot and its 256 registers), and takes a parked fit as soon as its record is published: eight lanes per
// fit, the arithmetic of coopfit8_kernel line for line (the trip body below is that kernel's, generated from the same text),
// as a state machine -- a row whose fit ends fetches the next record at the top of the following trip.
// Hand-off (cdna_hip_programming.md, Guideline 16, R1 with 8-byte agent atomics on both sides): the lane writes the record's
// first fourteen words with agent-scope 8-byte stores, waits for them (s_waitcnt vmcnt(0)), then the fifteenth -- nfev in the
// low half, the batch's TAG (never 0, unique per batch of a context: no stale record of an earlier batch can match) in the high
// half; a row claims an index only when the published count is ahead of the claim counter, polls that ONE word, and reads the
// record with agent-scope loads.  Nobody waits for anything that is not already on its way: a wave of this kernel leaves when the
// lane kernel's waves have all signed off (`done` == lane_waves), every published record is claimed and its own rows are idle -- or,
// if no record was EVER published, after `idle_us` of waiting (the two kernels were not run side by side: the lists then go to the
// cooperative launches behind the lane kernel, as before round 6; nothing was claimed, nothing is lost).
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) int gi32;
__device__ __forceinline__ unsigned long long ld_agent(const void* p) {
    return __hip_atomic_load((gu64*)(uintptr_t)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ld_agent_i(const int* p) {
    return __hip_atomic_load((gi32*)(uintptr_t)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(void* p, unsigned long long v) {
    __hip_atomic_store((gu64*)(uintptr_t)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double u2d(unsigned long long v) { return __longlong_as_double((long long)v); }
static_assert(sizeof(ParkedFit) == 120, "fifteen 8-byte words: the last one carries nfev and the tag");
// the lane kernel's side: publish one record
__device__ __forceinline__ void publish_parked(ParkedFit* slot, const ParkedFit& pf, unsigned tag) {
    const unsigned long long* w = reinterpret_cast<const unsigned long long*>(&pf);
#pragma unroll
    for (int i = 0; i < 14; ++i) st_agent(reinterpret_cast<unsigned long long*>(slot) + i, w[i]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st_agent(reinterpret_cast<unsigned long long*>(slot) + 14, (unsigned long long)(unsigned)pf.nfev | ((unsigned long long)tag << 32));
}

#ifdef MPX_DEV_KNOBS   // measured, not adopted (see the verdict below): development builds only
__global__ __launch_bounds__(256) void coopfit_live_kernel(const ParkedFit* parked, const int* parked_count, int* next_parked,
                                                         const int* lane_done, int lane_waves, unsigned tag,
                                                         const double* __restrict__ y, double* center, int* ok, int maxfev,
                                                         int* evals, int idle_us, int* live_stats, int poll_mask) {
    using namespace lm;
    __shared__ double exp_tab[64];
    if (threadIdx.x < 64) exp_tab[threadIdx.x] = exp2((double)threadIdx.x * (1.0 / 64.0));
    __syncthreads();
    const int l = threadIdx.x & 7;
    unsigned my_evals = 0, my_fits = 0;
    const double ftol = 1.49012e-8, xtol = 1.49012e-8, gtol = 0.0, factor = 100.0;
    const double eps = sqrt(EPSMCH);
    // per-row state (the registers coopfit8_kernel keeps across the trips of one fit)
    bool have = false, ona = false, onb = false, onc = false, need_outer = true;
    int pending = -1;   // an index claimed but not yet published
    long long outw = 0, row_off = 0;
    double x0w = 0.0;
    int mw = 0, nfev0 = 0;
    D3 px = {0, 0, 0}, py = {0, 0, 0}, f = {0, 0, 0};
    double x[NP] = {0, 0, 0}, diag[NP] = {1, 1, 1};
    double par = 0.0, delta = 0.0, xnorm = 0.0, fnorm = 0.0, gnorm = 0.0;
    int it = 1, nfev = 0;
    int ipvt[NP] = {0, 1, 2};
    double qtf[NP] = {0.0, 0.0, 0.0}, r[NP * NP];
#pragma unroll
    for (int i = 0; i < NP * NP; ++i) r[i] = 0.0;
    auto resid = [&](const double* p) -> D3 {
        const GaussEval g = gauss_prep(p, exp_tab);
        const double ra = gauss_resid(g, px.a, py.a), rb = gauss_resid(g, px.b, py.b), rc = gauss_resid(g, px.c, py.c);
        return {ona ? ra : 0.0, onb ? rb : 0.0, onc ? rc : 0.0};
    };
    __builtin_amdgcn_s_setprio(3);   // a wave of this kernel issues one instruction in ten clocks (a dependent chain): it goes first, the lane kernel's wave on the same SIMD keeps the other nine
    const long long t_start = wall_clock64();   // 100 MHz
    bool seen_any = false;   // wave-uniform: a record has been published in this batch
    int trip = 0;
    for (;;) {
        // ---------------- fetch: idle rows look for a published record (while other rows of the wave are at work: every
        // poll_mask + 1 trips only -- the agent-scope loads are a microsecond of latency the whole wave waits for)
        ++trip;
        const bool wave_at_work = __any(have);
        if (!have && ((trip & poll_mask) == 0 || !wave_at_work)) {
            int got = -1;
            if (l == 0) {
                if (pending < 0) {
                    const int nx = ld_agent_i(next_parked), c = ld_agent_i(parked_count);
                    if (nx < c) pending = atomicAdd(next_parked, 1);   // (may overshoot c by the rows racing here: such a claim waits for a later record, or for the end)
                }
                if (pending >= 0) {
                    const unsigned long long w14 = ld_agent(reinterpret_cast<const unsigned long long*>(parked + pending) + 14);
                    if ((unsigned)(w14 >> 32) == tag && pending < ld_agent_i(parked_count)) {
                        got = pending;
                        pending = -1;
                    }
                }
            }
            got = __shfl(got, 0, 8);
            if (got >= 0) {
                const unsigned long long* w = reinterpret_cast<const unsigned long long*>(parked + got);
                outw = (long long)ld_agent(w + 0);
                row_off = (long long)ld_agent(w + 1);
                x0w = u2d(ld_agent(w + 2));
                x[0] = u2d(ld_agent(w + 3));
                x[1] = u2d(ld_agent(w + 4));
                x[2] = u2d(ld_agent(w + 5));
                diag[0] = u2d(ld_agent(w + 6));
                diag[1] = u2d(ld_agent(w + 7));
                diag[2] = u2d(ld_agent(w + 8));
                par = u2d(ld_agent(w + 9));
                delta = u2d(ld_agent(w + 10));
                xnorm = u2d(ld_agent(w + 11));
                fnorm = u2d(ld_agent(w + 12));
                const unsigned long long w13 = ld_agent(w + 13), w14 = ld_agent(w + 14);
                mw = (int)(unsigned)w13;
                it = (int)(unsigned)(w13 >> 32);
                nfev = nfev0 = (int)(unsigned)w14;
                ona = l < mw;
                onb = l + 8 < mw;
                onc = l + 16 < mw;
                px = {x0w + (double)l, x0w + (double)(l + 8), x0w + (double)(l + 16)};
                py = {ona ? y[row_off + l] : 0.0, onb ? y[row_off + l + 8] : 0.0, onc ? y[row_off + l + 16] : 0.0};
                f = resid(x);
                need_outer = true;
                gnorm = 0.0;
                have = true;
            }
        }
        if (!__any(have)) {
            // nothing to do in this wave: leave when the lane kernel is done and every published record is claimed
            const int dn = ld_agent_i(lane_done);
            const int c = ld_agent_i(parked_count);   // (read AFTER `done`: a wave signs off after its last publication's count)
            seen_any = seen_any || c > 0;
            const bool mine = l == 0 && pending >= 0 && pending < c;   // a claim of ours that a record stands (or will stand) behind
            if (dn >= lane_waves && !__any(mine) && ld_agent_i(next_parked) >= c) break;
            if (!seen_any && dn == 0 && (wall_clock64() - t_start) > (long long)idle_us * 100) break;   // never ran side by side
            __builtin_amdgcn_s_sleep(24);
            continue;
        }
        int info = 0;
        if (have) {
            if (need_outer) {
            // forward-difference jacobian: this lane's three rows
                D3 J0, J1, J2;
                {
                    if (x[0] != 0.0) {
                        const double inv_a = lm_rcp(x[0]);
                        J0 = {ona ? (f.a + py.a) * inv_a : 0.0, onb ? (f.b + py.b) * inv_a : 0.0, onc ? (f.c + py.c) * inv_a : 0.0};
                    } else {
                        x[0] = eps;
                        const D3 w = resid(x);
                        J0 = {(w.a - f.a) * (1.0 / eps), (w.b - f.b) * (1.0 / eps), (w.c - f.c) * (1.0 / eps)};
                        x[0] = 0.0;
                    }
                    const FdStep fs = fd_prep(x, gauss_prep(x, exp_tab).ninv, eps);
                    if (!fd_big(fs, x0w - x[1], x0w + (double)(mw - 1) - x[1])) {
                        fd_row(fs, px.a - x[1], f.a + py.a, J1.a, J2.a);
                        fd_row(fs, px.b - x[1], f.b + py.b, J1.b, J2.b);
                        fd_row(fs, px.c - x[1], f.c + py.c, J1.c, J2.c);
                        J1 = {ona ? J1.a : 0.0, onb ? J1.b : 0.0, onc ? J1.c : 0.0};
                        J2 = {ona ? J2.a : 0.0, onb ? J2.b : 0.0, onc ? J2.c : 0.0};
                    } else {   // a row outside the series' range (a width far below a lag): the two exponentials per row
                        const double keep1 = x[1], keep2 = x[2];
                        x[1] = fs.mu1;
                        const D3 w1 = resid(x);
                        x[1] = keep1;
                        x[2] = keep2 + (eps * fabs(keep2) == 0.0 ? eps : eps * fabs(keep2));
                        const D3 w2 = resid(x);
                        x[2] = keep2;
                        J1 = {(w1.a - f.a) * fs.inv_h1, (w1.b - f.b) * fs.inv_h1, (w1.c - f.c) * fs.inv_h1};
                        J2 = {(w2.a - f.a) * fs.inv_h2, (w2.b - f.b) * fs.inv_h2, (w2.c - f.c) * fs.inv_h2};
                    }
                }
                nfev += NP;
                ipvt[0] = 0;
                ipvt[1] = 1;
                ipvt[2] = 2;
                double acnorm[NP], rdiag[NP], wa[NP];
                acnorm[0] = lm_sqrt(coop8_sum(l, 0, J0, J0));
                acnorm[1] = lm_sqrt(coop8_sum(l, 0, J1, J1));
                acnorm[2] = lm_sqrt(coop8_sum(l, 0, J2, J2));
#pragma unroll
                for (int j = 0; j < NP; ++j) rdiag[j] = wa[j] = acnorm[j];
                D3 w4 = f;  // becomes Q^T fvec
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    int kmax = j;
                    double rmax = rdiag[j];
#pragma unroll
                    for (int k = j + 1; k < NP; ++k)
                        if (rdiag[k] > rmax) {
                            kmax = k;
                            rmax = rdiag[k];
                        }
                    if (kmax != j) {
                        D3& cjs = j == 0 ? J0 : (j == 1 ? J1 : J2);
                        D3& cks = kmax == 1 ? J1 : J2;
                        const D3 t0 = cjs;
                        cjs = cks;
                        cks = t0;
                        put3(rdiag, kmax, rdiag[j]);
                        put3(wa, kmax, wa[j]);
                        const int t = ipvt[j];
                        ipvt[j] = sel3(ipvt, kmax);
                        put3(ipvt, kmax, t);
                    }
                    D3& cj = j == 0 ? J0 : (j == 1 ? J1 : J2);
                    const bool below = l >= j;  // rows j..7 of the first slot; the other two slots (rows 8.., 16..) are always below
                    double ajnorm = lm_sqrt(coop8_sum(l, j, cj, cj));
                    if (ajnorm != 0.0) {
                        if (half_bcast(cj.a, j) < 0.0) ajnorm = -ajnorm;
                        const double inv_aj = lm_rcp(ajnorm);
                        if (below) cj.a *= inv_aj;
                        cj.b *= inv_aj;
                        cj.c *= inv_aj;
                        if (l == j) cj.a += 1.0;
                        const double inv_ajj = lm_rcp(half_bcast(cj.a, j));
#pragma unroll
                        for (int k = j + 1; k < NP; ++k) {
                            D3& ck = k == 1 ? J1 : J2;
                            const double temp = coop8_sum(l, j, cj, ck) * inv_ajj;
                            if (below) ck.a = fma_as_written(-temp, cj.a, ck.a);
                            ck.b = fma_as_written(-temp, cj.b, ck.b);
                            ck.c = fma_as_written(-temp, cj.c, ck.c);
                            if (rdiag[k] != 0.0) {
                                const double t = lm_div(half_bcast(ck.a, j), rdiag[k]);
                                const double u = fma(-t, t, 1.0);
                                rdiag[k] *= lm_sqrt(u > 0.0 ? u : 0.0);
                                const double q = lm_div(rdiag[k], wa[k]);
                                if (0.05 * q * q <= EPSMCH) {
                                    rdiag[k] = lm_sqrt(coop8_sum(l, j + 1, ck, ck));
                                    wa[k] = rdiag[k];
                                }
                            }
                        }
                        const double temp = -coop8_sum(l, j, cj, w4) * inv_ajj;
                        if (below) w4.a = fma_as_written(cj.a, temp, w4.a);
                        w4.b = fma_as_written(cj.b, temp, w4.b);
                        w4.c = fma_as_written(cj.c, temp, w4.c);
                    }
                    rdiag[j] = -ajnorm;
                    qtf[j] = half_bcast(w4.a, j);
                }
                if (it == 1) {
                    double wa3[NP];
#pragma unroll
                    for (int j = 0; j < NP; ++j) {
                        diag[j] = acnorm[j] != 0.0 ? acnorm[j] : 1.0;
                        wa3[j] = diag[j] * x[j];
                    }
                    xnorm = enorm3(wa3);
                    delta = factor * xnorm;
                    if (delta == 0.0) delta = factor;
                }
                // replicate the 3x3 upper triangle R (row i lives in lane i of the fit's eight, first slot; its diagonal is rdiag)
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    r[i * NP + 0] = i == 0 ? rdiag[0] : half_bcast(J0.a, i);
                    r[i * NP + 1] = i == 1 ? rdiag[1] : half_bcast(J1.a, i);
                    r[i * NP + 2] = i == 2 ? rdiag[2] : half_bcast(J2.a, i);
                }
                gnorm = 0.0;
                if (fnorm != 0.0) {
#pragma unroll
                    for (int j = 0; j < NP; ++j) {
                        const double an = sel3(acnorm, ipvt[j]);
                        if (an != 0.0) {
                            double s2 = 0.0;
#pragma unroll
                            for (int i = 0; i <= j; ++i) s2 = fma(r[i * NP + j], lm_div(qtf[i], fnorm), s2);
                            const double g = fabs(lm_div(s2, an));
                            gnorm = g > gnorm ? g : gnorm;
                        }
                    }
                }
                if (gnorm <= gtol) {
                    info = 4;
                } else {
#pragma unroll
                    for (int j = 0; j < NP; ++j) diag[j] = diag[j] > acnorm[j] ? diag[j] : acnorm[j];
                }
            }
            if (info == 0) {
                double rr[NP * NP], p[NP], xnew[NP], wa3[NP], sd[NP];
#pragma unroll
                for (int i = 0; i < NP * NP; ++i) rr[i] = r[i];
                par = lmpar(rr, ipvt, diag, qtf, delta, par, p, sd);
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    p[j] = -p[j];
                    xnew[j] = x[j] + p[j];
                    wa3[j] = diag[j] * p[j];
                }
                const double pnorm = enorm3(wa3);
                if (it == 1) delta = delta < pnorm ? delta : pnorm;
                const D3 fn = resid(xnew);
                ++nfev;
                const double fnorm1 = lm_sqrt(coop8_sum(l, 0, fn, fn));
                double actred = -1.0;
                if (0.1 * fnorm1 < fnorm) {
                    const double q = lm_div(fnorm1, fnorm);
                    actred = fma(-q, q, 1.0);
                }
#pragma unroll
                for (int j = 0; j < NP; ++j) wa3[j] = 0.0;
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const double temp = sel3(p, ipvt[j]);
#pragma unroll
                    for (int i = 0; i <= j; ++i) wa3[i] = fma(r[i * NP + j], temp, wa3[i]);
                }
                const double temp1 = lm_div(enorm3(wa3), fnorm);
                const double temp2 = lm_div(lm_sqrt(par) * pnorm, fnorm);
                const double prered = fma(temp1, temp1, temp2 * temp2 / 0.5);
                const double dirder = -fma(temp1, temp1, temp2 * temp2);
                const double ratio = prered != 0.0 ? lm_div(actred, prered) : 0.0;
                if (ratio <= 0.25) {
                    double temp = actred >= 0.0 ? 0.5 : lm_div(0.5 * dirder, dirder + 0.5 * actred);
                    if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
                    const double p10 = lm_div(pnorm, 0.1), dm = delta < p10 ? delta : p10;
                    delta = temp * dm;
                    par = lm_div(par, temp);
                } else if (par == 0.0 || ratio >= 0.75) {
                    delta = pnorm / 0.5;
                    par = 0.5 * par;
                }
                if (ratio >= 1e-4) {
#pragma unroll
                    for (int j = 0; j < NP; ++j) {
                        x[j] = xnew[j];
                        wa3[j] = diag[j] * x[j];
                    }
                    f = fn;
                    xnorm = enorm3(wa3);
                    fnorm = fnorm1;
                    ++it;
                }
                const bool c1 = fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0;
                if (c1) info = 1;
                if (delta <= xtol * xnorm) info = 2;
                if (c1 && info == 2) info = 3;
                if (info == 0) {
                    if (nfev >= maxfev) info = 5;
                    if (fabs(actred) <= EPSMCH && prered <= EPSMCH && 0.5 * ratio <= 1.0) info = 6;
                    if (delta <= EPSMCH * xnorm) info = 7;
                    if (gnorm <= EPSMCH) info = 8;
                }
                need_outer = ratio >= 1e-4;   // accepted: a new jacobian next trip; rejected: the next trial with the same R
            }
            if (info != 0) {
                if (l == 0) {
                    ok[outw] = (info >= 1 && info <= 4) ? 1 : 0;
                    center[outw] = x[1];
                    my_evals += (unsigned)(nfev - nfev0);
                    ++my_fits;
                }
                have = false;
            }
        }
    }
    count_evals(evals, my_evals);   // total[6..7]
    if (live_stats) {
        unsigned v = my_fits;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(live_stats, (int)v);
    }
}
#endif

// SAMPLES_IN_LDS = true (batches with enough fits to fill the machine several times over): the lane's 21 samples live in
// LDS and fvec is recomputed -- no global loads inside the trip loop (10 % faster per 2.1 M fits, 1/31 of the traffic).
// false (small batches, where the kernel's time is the number of dependent trips of its longest fits, not throughput): the
// round-2 arrangement, fvec in LDS and the samples re-read from the ESACF row (L2) -- 21 exponentials less per trip.
// The two compute the SAME BITS (fvec recomputed is fvec stored), so the choice is pure scheduling.
template <bool SAMPLES_IN_LDS>
__global__ __launch_bounds__(FIT_THREADS, FIT_WAVES_PER_SIMD) void peakfit_kernel(
    int* total_peaks, int* next_item, const int* __restrict__ worklist,
    int shard_cap, const unsigned long long* __restrict__ shard_cnt, const double* __restrict__ y, int Mh, int maxp, const int* __restrict__ peak_idx,
    double* center, int* ok, int maxfev, ParkedFit* parked, int* parked_count, int park_nfev, int park_live, int park_cap,
    int early_nfev, int early_cap, unsigned live_tag, int* lane_done) {
    using namespace lm;
    __shared__ double sh[MAXM * FIT_THREADS];
    __shared__ double sh_rq[9 * FIT_THREADS];
    __shared__ double exp_tab[64];
    if (threadIdx.x < 64) exp_tab[threadIdx.x] = exp2((double)threadIdx.x * (1.0 / 64.0));
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // The 21 samples of the lane's fit live in LDS for the life of the fit (element i at ysl[i * 64]; rows >= m are zeros):
    // read from the ESACF row ONCE, at fetch time.  Until round 3 this space held MINPACK's fvec and the samples were
    // re-read from the row twice per trip -- 6.4 GB of L2-miss traffic per 2.1 M fits for 0.38 GB of windows, 64 different
    // cache lines per load instruction, and a wait in front of every evaluation.  fvec is not stored at all now: it is
    // the residual at the current x, and the OUTER section recomputes it row by row next to the two jacobian evaluations
    // (the same function of the same inputs: the same bits; three independent exponentials per row instead of two).
    double* ysl = sh + (size_t)wave * MAXM * 64 + lane;
    double* fvec = ysl;   // SAMPLES_IN_LDS == false: the same space holds fvec (element i at fvec[i * 64])
    double* rq = sh_rq + (size_t)wave * 9 * 64 + lane;      // R (6) and Q^T f (3) between OUTER and INNER
    // The work list is WL_SHARDS regions (peak_pick): item number wi of the kernel's one running counter is long fit wi of the
    // concatenated shards' long fits, or -- past all of those -- other fit wi - n_long of the concatenated others.  Prefix sums of
    // the shards' counts in LDS (every workgroup its own copy), a five-step search per fetched item.
    __shared__ int wl_long_pre[WL_SHARDS + 1], wl_other_pre[WL_SHARDS + 1];
    if (threadIdx.x < 64) {
        const unsigned long long c = threadIdx.x < WL_SHARDS ? shard_cnt[threadIdx.x * WL_SHARD_STRIDE] : 0ull;
        int pl = (int)(unsigned)c, po = (int)(c >> 32);
#pragma unroll
        for (int off = 1; off < WL_SHARDS; off <<= 1) {   // inclusive scans over the lanes 0 .. WL_SHARDS - 1
            const int ql = __shfl_up(pl, off), qo = __shfl_up(po, off);
            if ((int)threadIdx.x >= off) {
                pl += ql;
                po += qo;
            }
        }
        if (threadIdx.x < WL_SHARDS) {
            wl_long_pre[threadIdx.x + 1] = pl;
            wl_other_pre[threadIdx.x + 1] = po;
        }
        if (threadIdx.x == 0) wl_long_pre[0] = wl_other_pre[0] = 0;
    }
    __syncthreads();
    const int n_long = wl_long_pre[WL_SHARDS], total = n_long + wl_other_pre[WL_SHARDS];
    if (blockIdx.x == 0 && threadIdx.x == 0) {   // the batch's totals where the host and the statistics read them
        total_peaks[0] = n_long;
        total_peaks[2] = total - n_long;
    }
    auto wl_item = [&](int wi) -> int {
        const bool is_long = wi < n_long;
        const int k = is_long ? wi : wi - n_long;
        const int* pre = is_long ? wl_long_pre : wl_other_pre;
        int s = 0;
#pragma unroll
        for (int step = WL_SHARDS / 2; step >= 1; step >>= 1)
            if (pre[s + step] <= k) s += step;          // the last shard whose prefix is <= k
        const int r = k - pre[s];
        return worklist[is_long ? s * shard_cap + r : (s + 1) * shard_cap - 1 - r];
    };
    const double ftol = 1.49012e-8, xtol = 1.49012e-8, gtol = 0.0, factor = 100.0;
    const double eps = sqrt(EPSMCH);

    int phase = FIT_NEED_WORK;
    unsigned my_evals = 0;   // statistics: MINPACK function evaluations this lane has run
    bool drained = false;  // wave-uniform: some lane has found the work list empty
    bool cap_hit = false;  // wave-uniform: the cooperative kernel is full, park only from a thinned-out wave
    bool early_full = false;   // this lane has seen the early-parking budget of the batch used up
    // per-fit state
    const double* row = y;
    double x0 = 0.0, x[NP] = {0, 0, 0}, diag[NP] = {1, 1, 1};
    double par = 0.0, delta = 0.0, xnorm = 0.0, fnorm = 0.0, gnorm = 0.0;
    int ipvt[NP] = {0, 1, 2};
    int m = 0, nfev = 0, it = 1;
    long long out = 0;

    for (;;) {
        // ---------------- fetch
        if (phase == FIT_NEED_WORK) {
            const unsigned long long need = __ballot(1);  // lanes inside this branch
            const int cnt = __popcll(need);
            const int rank = __popcll(need & ((1ull << lane) - 1ull));
            int base = 0;
            if (rank == 0) base = atomicAdd(next_item, cnt);
            base = __shfl(base, __ffsll((long long)need) - 1);
            const int wi = base + rank;
            if (wi >= total) {
                phase = FIT_DONE;
            } else {
                const int item = wl_item(wi);
                const long long f = item >> 12;
                const int j = item & 0xfff;
                const int i = peak_idx[f * maxp + j];
                out = f * maxp + j;
                const int stop = i + 11 < Mh ? i + 11 : Mh;
                m = stop - (i - 10);
                // slice(i-10, i+11): a negative start wraps in Python -> empty slice -> the fit raises -> dropped
                if (i < 10 || m < 3) {
                    ok[out] = 0;  // stays in NEED_WORK: fetches again on the next trip
                } else {
                    row = y + f * (long long)Mh + (i - 10);
                    x0 = (double)(i - 10);
                    double ys0[MAXM];
                    load_samples(row, m, ys0);
                    if (SAMPLES_IN_LDS) {
#pragma unroll
                        for (int q = 0; q < MAXM; ++q) ysl[q * 64] = ys0[q];
                    }
                    double ymax = ys0[0];
#pragma unroll
                    for (int q = 1; q < MAXM; ++q)
                        if (q < m) ymax = ys0[q] > ymax ? ys0[q] : ymax;
                    x[0] = ymax;  // peakutils initial guess: [max(y), x[0], 5*(x[1]-x[0])]
                    x[1] = x0;
                    x[2] = 5.0;
                    phase = FIT_INIT;  // the residuals at the initial point come from the shared evaluation below
                }
            }
        }
        // End game: once the work list is empty, a fit that is already past PARK_NFEV evaluations (a runaway fit:
        // the others need ~60) would keep this wave -- and the kernel -- alive for up to 200 more trips of
        // ~45 us.  Its MINPACK state is handed to coopfit_kernel instead, which finishes all such fits at once
        // with 16 lanes each.
        drained = drained || __any(phase == FIT_DONE);
        // ... as long as coopfit_kernel can hold them all at once (park_cap), and beyond that only from waves that
        // have thinned out: a wave that still holds many runaway fits (short frames: 11 % of the fits at N = 1023)
        // runs them at good lane utilisation, far cheaper than 16 lanes per fit.  The cap is enforced with the return
        // value of an atomic on its own counter (reading the hot parked counter every trip stalled every wave).
        const int live = __popcll(__ballot(phase != FIT_DONE));
        bool may_park = drained && phase == FIT_OUTER && nfev >= park_nfev && parked != nullptr;
        if (may_park && live > park_live) {
            may_park = !cap_hit && atomicAdd(parked_count + 2, 1) < park_cap;
            cap_hit = cap_hit || __any(!may_park);
        }
        // Round 6, development builds (early_nfev > 0; measured, not adopted -- with or without coopfit_live_kernel next to this
        // kernel the call is slower, profiles/r6/fit_early_sweep.txt): a fit that LOOKS like a runaway -- its width or the distance of its centre from
        // the peak has left everything a converging fit visits (40 lags; oracle statistics over 10 299 fits of the Target's
        // signal: every fit that burns maxfev shows it by evaluation 93, 90 % by 41, and 5 % of the others, slow ones, do too)
        // -- is handed to the cooperative kernel at once, list drained or not: a wave no longer carries such lanes for forty
        // trips, and the kernel ends when its ordinary fits do.  Which kernel runs which evaluation is pure scheduling (one
        // arithmetic, see above): the bits do not depend on it.  Bounded by a counter of its own.
        bool early_park = false;   // parked with the list not drained: the lane fetches its next fit
        if (early_nfev > 0 && !may_park && phase == FIT_OUTER && nfev >= early_nfev && parked != nullptr && !early_full &&
            (fabs(x[2]) > 40.0 || fabs(x[1] - (x0 + 10.0)) > 40.0 || nfev >= EARLY_PARK_SLOW_NFEV)) {
            may_park = atomicAdd(parked_count + 7, 1) < early_cap;
            early_full = !may_park;
            early_park = may_park;
        }
        if (may_park) {
            ParkedFit pf;
            pf.out = out;
            pf.row_off = row - y;
            pf.x0 = x0;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                pf.x[j] = x[j];
                pf.diag[j] = diag[j];
            }
            pf.par = par;
            pf.delta = delta;
            pf.xnorm = xnorm;
            pf.fnorm = fnorm;
            pf.m = m;
            pf.it = it;
            pf.nfev = nfev;
            pf.pad = 0;
            if (live_tag)   // coopfit_live_kernel is running next to this kernel: the record is published word by word, its tag last
                publish_parked(parked + atomicAdd(parked_count, 1), pf, live_tag);
            else
                parked[atomicAdd(parked_count, 1)] = pf;
            my_evals += (unsigned)nfev;
            phase = early_park ? FIT_NEED_WORK : FIT_DONE;
        }
        if (__all(phase == FIT_DONE)) break;

        int info = 0;
#if defined(MPX_DEV_KNOBS) && defined(MPX_FIT_STATS)
        if (lane == 0) FIT_STAT(18, 1);
        {
            const int no = __popcll(__ballot(phase == FIT_OUTER));
            if (lane == 0 && no) {
                FIT_STAT(16, 1);
                FIT_STAT(17, no);
            }
        }
#endif
        // ---------------- OUTER: jacobian, QR, Q^T f
        if (phase == FIT_OUTER) {
            double a[MAXM][NP], r[NP * NP], qtf[NP];
            double w[MAXM];  // fvec (the residuals at x) here, Q^T fvec after the factorisation
            {
                // column 0 (amplitude): ((A+h) e_i - y_i - f_i)/h with f_i = A e_i - y_i is e_i itself up to the
                // rounding noise of the difference; e_i = (f_i + y_i)/A is read off the residuals instead
                // of 21 more exponentials (exact path kept for A == 0)
                double ys_g[MAXM];
                if (!SAMPLES_IN_LDS) load_samples(row, m, ys_g);
                const GaussEval g0 = gauss_prep(x, exp_tab);   // (dead code without SAMPLES_IN_LDS)
                const bool a_nonzero = x[0] != 0.0;
                const double inv_a = a_nonzero ? lm_rcp(x[0]) : 0.0;
                double xa[NP] = {eps, x[1], x[2]};
                const GaussEval ga = gauss_prep(xa, exp_tab);   // only used when A == 0
                const FdStep fs = fd_prep(x, g0.ninv, eps);
                const double inv_ha = 1.0 / eps;
                // (rare: a width far below a lag) this fit takes the two exponentials per row, in place: a branch per row that a
                // wave without such a fit skips
                const bool big = fd_big(fs, x0 - x[1], x0 + (double)(m - 1) - x[1]);
                const GaussEval g1 = {x[0], fs.mu1, g0.ninv, exp_tab}, g2 = {x[0], x[1], fs.ninv2, exp_tab};
#pragma unroll
                for (int i = 0; i < MAXM; ++i) {
                    const double yi = SAMPLES_IN_LDS ? ysl[i * 64] : ys_g[i], xi = x0 + (double)i;
                    const double fi = SAMPLES_IN_LDS ? gauss_resid(g0, xi, yi) : fvec[i * 64];
                    if (a_nonzero)
                        a[i][0] = keep_if(i < m, (fi + yi) * inv_a);
                    else
                        a[i][0] = i < m ? (gauss_resid(ga, xi, yi) - fi) * inv_ha : 0.0;
                    double j1, j2;
                    fd_row(fs, xi - x[1], fi + yi, j1, j2);
                    if (big) {
                        j1 = (gauss_resid(g1, xi, yi) - fi) * fs.inv_h1;
                        j2 = (gauss_resid(g2, xi, yi) - fi) * fs.inv_h2;
                    }
                    a[i][1] = keep_if(i < m, j1);
                    a[i][2] = keep_if(i < m, j2);
                    w[i] = keep_if(i < m, fi);
                }
            }
            nfev += NP;
            ipvt[0] = 0;
            ipvt[1] = 1;
            ipvt[2] = 2;
            double acnorm[NP], rdiag[NP], wa[NP];
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const double q = dot_rows(0, [&](int i) { return a[i][j]; }, [&](int i) { return a[i][j]; });
                acnorm[j] = lm_sqrt(q);
                rdiag[j] = wa[j] = acnorm[j];
            }
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                int kmax = j;
                double rmax = rdiag[j];
#pragma unroll
                for (int k = j + 1; k < NP; ++k)
                    if (rdiag[k] > rmax) {
                        kmax = k;
                        rmax = rdiag[k];
                    }
                if (kmax != j) {
#pragma unroll
                    for (int i = 0; i < MAXM; ++i) {
                        const double t = a[i][j];
                        const double o = kmax == 1 ? a[i][1] : a[i][2];
                        a[i][j] = o;
                        if (kmax == 1) a[i][1] = t;
                        else a[i][2] = t;
                    }
                    put3(rdiag, kmax, rdiag[j]);
                    put3(wa, kmax, wa[j]);
                    const int t = ipvt[j];
                    ipvt[j] = sel3(ipvt, kmax);
                    put3(ipvt, kmax, t);
                }
                const double q = dot_rows(j, [&](int i) { return a[i][j]; }, [&](int i) { return a[i][j]; });
                double ajnorm = lm_sqrt(q);
                if (ajnorm != 0.0) {
                    if (a[j][j] < 0.0) ajnorm = -ajnorm;
                    const double inv_aj = lm_rcp(ajnorm);
#pragma unroll
                    for (int i = j; i < MAXM; ++i) a[i][j] *= inv_aj;
                    a[j][j] += 1.0;
                    const double inv_ajj = lm_rcp(a[j][j]);
#pragma unroll
                    for (int k = j + 1; k < NP; ++k) {
                        const double sum = dot_rows(j, [&](int i) { return a[i][j]; }, [&](int i) { return a[i][k]; });
                        const double temp = sum * inv_ajj;
#pragma unroll
                        for (int i = j; i < MAXM; ++i) a[i][k] = fma_as_written(-temp, a[i][j], a[i][k]);
                        if (rdiag[k] != 0.0) {
                            const double t = lm_div(a[j][k], rdiag[k]);
                            const double u = fma(-t, t, 1.0);
                            rdiag[k] *= lm_sqrt(u > 0.0 ? u : 0.0);
                            const double qq = lm_div(rdiag[k], wa[k]);
                            if (0.05 * qq * qq <= EPSMCH) {
                                const double s2 = dot_rows(j + 1, [&](int i) { return a[i][k]; }, [&](int i) { return a[i][k]; });
                                rdiag[k] = lm_sqrt(s2);
                                wa[k] = rdiag[k];
                            }
                        }
                    }
                    // the same reflection applied to the copy of fvec (MINPACK does this after qrfac; rows >= m
                    // of both the vector and the column are zero)
                    {
                        const double sum = dot_rows(j, [&](int i) { return a[i][j]; }, [&](int i) { return w[i]; });
                        const double temp = -sum * inv_ajj;
#pragma unroll
                        for (int i = j; i < MAXM; ++i) w[i] = fma_as_written(a[i][j], temp, w[i]);
                    }
                }
                rdiag[j] = -ajnorm;
                qtf[j] = w[j];
            }
            if (it == 1) {
                double wa3[NP];
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    diag[j] = acnorm[j] != 0.0 ? acnorm[j] : 1.0;
                    wa3[j] = diag[j] * x[j];
                }
                xnorm = enorm3(wa3);
                delta = factor * xnorm;
                if (delta == 0.0) delta = factor;
            }
#pragma unroll
            for (int i = 0; i < NP; ++i)
#pragma unroll
                for (int j = 0; j < NP; ++j) r[i * NP + j] = i == j ? rdiag[j] : a[i][j];
            gnorm = 0.0;
            if (fnorm != 0.0) {
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const double an = sel3(acnorm, ipvt[j]);
                    if (an != 0.0) {
                        double s2 = 0.0;
#pragma unroll
                        for (int i = 0; i <= j; ++i) s2 = fma(r[i * NP + j], lm_div(qtf[i], fnorm), s2);
                        const double g = fabs(lm_div(s2, an));
                        gnorm = g > gnorm ? g : gnorm;
                    }
                }
            }
            // R's upper triangle and Q^T f wait in LDS for the INNER steps (they must survive the next OUTER section
            // of lanes that only retry a step, and 9 more doubles in registers there are 9 spilled ones)
            rq[0 * 64] = r[0];
            rq[1 * 64] = r[1];
            rq[2 * 64] = r[2];
            rq[3 * 64] = r[4];
            rq[4 * 64] = r[5];
            rq[5 * 64] = r[8];
#pragma unroll
            for (int j = 0; j < NP; ++j) rq[(6 + j) * 64] = qtf[j];
            if (gnorm <= gtol) {
                info = 4;
            } else {
#pragma unroll
                for (int j = 0; j < NP; ++j) diag[j] = diag[j] > acnorm[j] ? diag[j] : acnorm[j];
                phase = FIT_INNER;
            }
        }
        // ---------------- INNER: one trust-region trial
        if ((phase == FIT_INNER && info == 0) || phase == FIT_INIT) {
            const bool fresh = phase == FIT_INIT;
            double p[NP] = {0.0, 0.0, 0.0}, xnew[NP], wa3[NP];
            double pnorm = 0.0;
            double r[NP * NP], qtf[NP];
            r[0] = rq[0 * 64];
            r[1] = rq[1 * 64];
            r[2] = rq[2 * 64];
            r[4] = rq[3 * 64];
            r[5] = rq[4 * 64];
            r[8] = rq[5 * 64];
            r[3] = r[6] = r[7] = 0.0;  // below the diagonal: scratch of lmpar / qrsolv, never read
#pragma unroll
            for (int j = 0; j < NP; ++j) qtf[j] = rq[(6 + j) * 64];
#if defined(MPX_DEV_KNOBS) && defined(MPX_FIT_STATS)
            int lm_it = 11;
#endif
            if (!fresh) {
                double rr[NP * NP], sd[NP];
#pragma unroll
                for (int i = 0; i < NP * NP; ++i) rr[i] = r[i];
#if defined(MPX_DEV_KNOBS) && defined(MPX_FIT_STATS)
                par = lmpar(rr, ipvt, diag, qtf, delta, par, p, sd, &lm_it);
#else
                par = lmpar(rr, ipvt, diag, qtf, delta, par, p, sd);
#endif
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    p[j] = -p[j];
                    wa3[j] = diag[j] * p[j];
                }
                pnorm = enorm3(wa3);
                if (it == 1) delta = delta < pnorm ? delta : pnorm;
            }
#pragma unroll
            for (int j = 0; j < NP; ++j) xnew[j] = x[j] + p[j];
#if defined(MPX_DEV_KNOBS) && defined(MPX_FIT_STATS)
            {
                FIT_STAT(lm_it, 1);
                const int its = lm_it == 11 ? 0 : lm_it;
                int mx = its, sm = its;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const int o = __shfl_xor(mx, off), q = __shfl_xor(sm, off);   // (inactive lanes return their own value: see below)
                    mx = o > mx ? o : mx;
                    sm += q;
                }
                const unsigned long long act = __ballot(1);
                if (lane == __ffsll((long long)act) - 1) {
                    FIT_STAT(12, 1);
                    FIT_STAT(13, __popcll(act));
                }
                FIT_STAT(15, its);
                // wave maximum over the ACTIVE lanes: by atomicMax into LDS would be exact; the shuffle above reads inactive
                // lanes' stale registers, so take the maximum through a ballot per iteration count instead
                int wmax = 0;
                for (int k = 1; k <= 10; ++k)
                    if (__ballot(its >= k)) wmax = k;
                if (lane == __ffsll((long long)act) - 1) FIT_STAT(14, wmax);
                (void)mx;
                (void)sm;
            }
#endif
            double rn[MAXM];  // residuals at the trial point (MINPACK's wa4)
            {
                double ys_g[MAXM];
                if (!SAMPLES_IN_LDS) load_samples(row, m, ys_g);
                const GaussEval g = gauss_prep(xnew, exp_tab);
#pragma unroll
                for (int i = 0; i < MAXM; ++i)
                    rn[i] = keep_if(i < m, gauss_resid(g, x0 + (double)i, SAMPLES_IN_LDS ? ysl[i * 64] : ys_g[i]));
            }
            const double s1 = dot_rows(0, [&](int i) { return rn[i]; }, [&](int i) { return rn[i]; });
            if (fresh) {  // lmdif's prologue: fvec at the initial point and its norm
                if (!SAMPLES_IN_LDS) {
#pragma unroll
                    for (int i = 0; i < MAXM; ++i)
                        if (i < m) fvec[i * 64] = rn[i];
                }
                nfev = 1;
                fnorm = lm_sqrt(s1);
                par = 0.0;
                it = 1;
                diag[0] = diag[1] = diag[2] = 1.0;
                delta = xnorm = 0.0;
                phase = FIT_OUTER;
            } else {
            ++nfev;
            const double fnorm1 = lm_sqrt(s1);
            double actred = -1.0;
            if (0.1 * fnorm1 < fnorm) {
                const double q = lm_div(fnorm1, fnorm);
                actred = fma(-q, q, 1.0);
            }
#pragma unroll
            for (int j = 0; j < NP; ++j) wa3[j] = 0.0;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const double temp = sel3(p, ipvt[j]);
#pragma unroll
                for (int i = 0; i <= j; ++i) wa3[i] = fma(r[i * NP + j], temp, wa3[i]);
            }
            const double temp1 = lm_div(enorm3(wa3), fnorm);
            const double temp2 = lm_div(lm_sqrt(par) * pnorm, fnorm);
            const double prered = fma(temp1, temp1, temp2 * temp2 / 0.5);
            const double dirder = -fma(temp1, temp1, temp2 * temp2);
            const double ratio = prered != 0.0 ? lm_div(actred, prered) : 0.0;
            if (ratio <= 0.25) {
                double temp = actred >= 0.0 ? 0.5 : lm_div(0.5 * dirder, dirder + 0.5 * actred);
                if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
                const double p10 = lm_div(pnorm, 0.1), dm = delta < p10 ? delta : p10;
                delta = temp * dm;
                par = lm_div(par, temp);
            } else if (par == 0.0 || ratio >= 0.75) {
                delta = pnorm / 0.5;
                par = 0.5 * par;
            }
            if (ratio >= 1e-4) {
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    x[j] = xnew[j];
                    wa3[j] = diag[j] * x[j];
                }
                if (!SAMPLES_IN_LDS) {
#pragma unroll
                    for (int i = 0; i < MAXM; ++i)
                        if (i < m) fvec[i * 64] = rn[i];
                }
                xnorm = enorm3(wa3);
                fnorm = fnorm1;
                ++it;
            }
            const bool c1 = fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0;
            if (c1) info = 1;
            if (delta <= xtol * xnorm) info = 2;
            if (c1 && info == 2) info = 3;
            if (info == 0) {
                if (nfev >= maxfev) info = 5;
                if (fabs(actred) <= EPSMCH && prered <= EPSMCH && 0.5 * ratio <= 1.0) info = 6;
                if (delta <= EPSMCH * xnorm) info = 7;
                if (gnorm <= EPSMCH) info = 8;
            }
            if (info == 0 && ratio >= 1e-4) phase = FIT_OUTER;  // step accepted: new jacobian next trip
            }
        }
        if (info != 0) {
            ok[out] = (info >= 1 && info <= 4) ? 1 : 0;
            center[out] = x[1];
            my_evals += (unsigned)nfev;
            phase = FIT_NEED_WORK;
        }
    }
    count_evals(parked_count + 3, my_evals);   // total[6..7]
    // this wave will publish nothing more (every count it added was returned to it before this line): coopfit_live_kernel's
    // waves leave when all of the lane kernel's waves have signed off
    if (lane_done != nullptr && lane == 0) atomicAdd(lane_done, 1);
}

#pragma clang fp contract(fast)

#if defined(MPX_DEV_KNOBS) && defined(MPX_FIT_STATS)
}  // namespace mpx
// statistics builds only: read (and clear) the fit kernels' statistics, 64 counters (see g_fit_hist)
extern "C" int mpx_dev_fit_hist(unsigned long long* out64, int clear) {
    if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(mpx::g_fit_hist), 64 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (clear) {
        unsigned long long z[64] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(mpx::g_fit_hist), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
namespace mpx {
#endif

// ------------------------------------------------------------------ kernel 4
__global__ __launch_bounds__(64) void scatter_kernel(long long frame0, long long num_frames, int fs, int Mh, int maxp,
                                                     int note_names, const double* __restrict__ y,
                                                     const int* __restrict__ peak_count,
                                                     const int* __restrict__ peak_idx,
                                                     const double* __restrict__ center,
                                                     const int* __restrict__ ok, double* chroma_frames) {
    const long long lf = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (lf >= num_frames) return;
    double chroma[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) chroma[i] = 0.0;
    const int cnt = peak_count[lf];
    const double* row = y + lf * (long long)Mh;
    int s = 0;  // index among the fits that succeeded: pairs with peak_idx[s] (quirk A.8)
    for (int j = 0; j < cnt; ++j) {
        if (!ok[lf * maxp + j]) continue;
        const double tau = center[lf * maxp + j];
        const double weight = row[peak_idx[lf * maxp + s]];
        ++s;
        const double pitch = (double)fs / tau;
        const double midi = 12.0 * (log2(pitch) - log2(440.0)) + 69.0;
        if (!(midi == midi) || isinf(midi)) continue;  // NaN -> ValueError -> skipped (esacf.py:70-71)
        const long long note = (long long)nearbyint(midi);
        const int pc = (int)(((note % 12) + 12) % 12);
        // unicode-sharp quirk (A.18, MPX_NOTES_UNICODE): C#, D#, F#, G#, A# land in a stray dict key and are lost;
        // with ASCII note names (librosa < 0.8) every pitch class accumulates
        if (note_names == MPX_NOTES_UNICODE && (pc == 1 || pc == 3 || pc == 6 || pc == 8 || pc == 10)) continue;
#pragma unroll
        for (int q = 0; q < 12; ++q)
            if (q == pc) chroma[q] += weight;
    }
    double* out = chroma_frames + (frame0 + lf) * 12;
#pragma unroll
    for (int i = 0; i < 12; ++i) out[i] = chroma[i];
}

// ------------------------------------------------------------------ host side
static void host_fft(std::vector<cx<double>>& a) {  // in-place radix-2, forward; plan tables only
    const size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const long double ang = -2.0L * M_PIl * (long double)k / (long double)len;
                const cx<double> w = {(double)cosl(ang), (double)sinl(ang)};
                const cx<double> u = a[i + k];
                const cx<double> t = a[i + k + len / 2];
                const cx<double> v = {t.x * w.x - t.y * w.y, t.x * w.y + t.y * w.x};
                a[i + k] = {u.x + v.x, u.y + v.y};
                a[i + k + len / 2] = {u.x - v.x, u.y - v.y};
            }
    }
}

// frequency held by register e of thread t after dif_fft_keep_last<L>
template <int L>
static int dif_reg_freq_t(int t, int e) {
    constexpr int RL = DifPlan<L>::radix(DifPlan<L>::n - 1);
    return dif_freq<L>(dif_last_pos<L>(t, e / RL, e % RL));
}
static int dif_reg_freq(int L, int t, int e) {
    switch (L) {
        case 512: return dif_reg_freq_t<512>(t, e);
        case 1024: return dif_reg_freq_t<1024>(t, e);
        case 2048: return dif_reg_freq_t<2048>(t, e);
        default: return dif_reg_freq_t<4096>(t, e);
    }
}

struct EsacfPlan {
    cx<double>* tw = nullptr;
    cx<double>* chirp = nullptr;
    cx<double>* bhat = nullptr;
    int L = 0;
    bool blue = false;
};

static int esacf_plan(mpx_ctx* ctx, int N, EsacfPlan& plan, bool force_blue = false) {
    const std::string key = (force_blue ? "esacf_bN" : "esacf_N") + std::to_string(N);
    auto it = ctx->misc_plans.find(key);
    const bool pow2 = !force_blue && (N & (N - 1)) == 0 && N >= 512;  // smaller frames ride the 512-point Bluestein
    int L = 512;
    if (pow2)
        L = N;
    else
        while (L < 2 * N - 1) L <<= 1;
    plan.L = L;
    plan.blue = !pow2;
    if (it != ctx->misc_plans.end()) {
        plan.tw = (cx<double>*)it->second[0];
        plan.chirp = (cx<double>*)it->second[1];
        plan.bhat = (cx<double>*)it->second[2];
        return MPX_OK;
    }
    std::vector<cx<double>> tw(L);
    for (int j = 0; j < L; ++j) {
        const long double ang = -2.0L * M_PIl * j / (long double)L;
        tw[j] = {(double)cosl(ang), (double)sinl(ang)};
    }
    plan.tw = (cx<double>*)upload(ctx, tw.data(), tw.size() * sizeof(cx<double>));
    if (!plan.tw) return MPX_ENOMEM;
    if (!pow2) {
        std::vector<cx<double>> chirp(N), filt(L, cx<double>{0.0, 0.0});
        for (long long n = 0; n < N; ++n) {
            const long long q = (n * n) % (2LL * N);  // exact phase reduction
            const long double ang = M_PIl * (long double)q / (long double)N;
            chirp[n] = {(double)cosl(ang), (double)sinl(ang)};
        }
        filt[0] = chirp[0];
        for (int m = 1; m < N; ++m) filt[m] = filt[L - m] = chirp[m];
        host_fft(filt);
        for (auto& v : filt) {
            v.x /= L;
            v.y /= L;
        }
        // the kernel multiplies the filter spectrum onto registers that hold the DIF output of thread tid,
        // element e: store it in exactly that order ([e][tid], coalesced)
        std::vector<cx<double>> filt_regs(L);
        if (L > 4096)
            filt_regs = filt;  // sacf_big_kernel (Stockham engine) multiplies in natural order
        else
            for (int t = 0; t < L / 8; ++t)
                for (int e = 0; e < 8; ++e) filt_regs[(size_t)e * (L / 8) + t] = filt[dif_reg_freq(L, t, e)];
        plan.chirp = (cx<double>*)upload(ctx, chirp.data(), chirp.size() * sizeof(cx<double>));
        plan.bhat = (cx<double>*)upload(ctx, filt_regs.data(), filt_regs.size() * sizeof(cx<double>));
        if (!plan.chirp || !plan.bhat) return MPX_ENOMEM;
    }
    ctx->misc_plans[key] = {plan.tw, plan.chirp, plan.bhat};
    return MPX_OK;
}

// Tables of the prime-factor SACF engine for N = A0 * 3 * 11 * 31 (A0 = 1, 2), cached per N.
struct PfaPlan {
    const unsigned short* pos = nullptr;
    const unsigned short* pairs = nullptr;
    const cx<double>* cs31 = nullptr;
    const cx<double>* cs11 = nullptr;
    int npairs = 0, a0 = 0;
};

static bool pfa_supported(int N) { return N == 3 * 11 * 31 || N == 2 * 3 * 11 * 31; }

static int pfa_plan(mpx_ctx* ctx, int N, PfaPlan& plan) {
    const std::string key = "pfa_N" + std::to_string(N);
    const int A0 = N / (3 * 11 * 31);
    plan.a0 = A0;
    plan.npairs = N / 2 + 1;
    auto it = ctx->misc_plans.find(key);
    if (it != ctx->misc_plans.end()) {
        plan.pos = (const unsigned short*)it->second[0];
        plan.pairs = (const unsigned short*)it->second[1];
        plan.cs31 = (const cx<double>*)it->second[2];
        plan.cs11 = (const cx<double>*)it->second[3];
        return MPX_OK;
    }
    const int dims[4] = {A0, 3, 11, 31};
    auto position = [&](const int* r) { return ((r[0] * 3 + r[1]) * 11 + r[2]) * 31 + r[3]; };
    // index n = sum_i n_i (N / N_i) mod N  <=>  n_i = n * (N / N_i)^-1 mod N_i
    int inv[4];
    for (int i = 0; i < 4; ++i) {
        inv[i] = 0;
        for (int c = 0; c < dims[i]; ++c)
            if (((long long)(N / dims[i]) * c) % dims[i] == 1 % dims[i]) inv[i] = c;
    }
    std::vector<unsigned short> pos(N), pairs;
    for (int n = 0; n < N; ++n) {
        int r[4];
        for (int i = 0; i < 4; ++i) r[i] = (int)(((long long)(n % dims[i]) * inv[i]) % dims[i]);
        pos[n] = (unsigned short)position(r);
    }
    // mirror bin of the residue-indexed spectrum: position (k_i) <-> position (-k_i mod N_i)
    for (int p = 0; p < N; ++p) {
        int r[4], m[4], q = p;
        r[3] = q % 31; q /= 31;
        r[2] = q % 11; q /= 11;
        r[1] = q % 3;  q /= 3;
        r[0] = q;
        for (int i = 0; i < 4; ++i) m[i] = (dims[i] - r[i]) % dims[i];
        const int mp = position(m);
        if (p <= mp) {
            pairs.push_back((unsigned short)p);
            pairs.push_back((unsigned short)mp);
        }
    }
    if ((int)pairs.size() != 2 * plan.npairs) return set_error(ctx, MPX_EHIP, "internal: PFA pair table (%zu)", pairs.size());
    auto table = [](int P, int K, int H) {
        std::vector<cx<double>> t((size_t)K * H);
        for (int k = 0; k < K; ++k)
            for (int n = 1; n <= H; ++n) {
                const long double ang = 2.0L * M_PIl * (long double)((n * k) % P) / (long double)P;
                t[(size_t)k * H + n - 1] = {(double)cosl(ang), (double)sinl(ang)};
            }
        return t;
    };
    std::vector<cx<double>> t31(16 * 16);   // [k][n], n = 0..15: (cos, sin)(2 pi n k / 31)
    for (int k = 0; k < 16; ++k)
        for (int n = 0; n < 16; ++n) {
            const long double ang = 2.0L * M_PIl * (long double)((n * k) % 31) / 31.0L;
            t31[(size_t)k * 16 + n] = {(double)cosl(ang), (double)sinl(ang)};
        }
    const auto t11 = table(11, 6, 5);
    plan.pos = (const unsigned short*)upload(ctx, pos.data(), pos.size() * sizeof(unsigned short));
    plan.pairs = (const unsigned short*)upload(ctx, pairs.data(), pairs.size() * sizeof(unsigned short));
    plan.cs31 = (const cx<double>*)upload(ctx, t31.data(), t31.size() * sizeof(cx<double>));
    plan.cs11 = (const cx<double>*)upload(ctx, t11.data(), t11.size() * sizeof(cx<double>));
    if (!plan.pos || !plan.pairs || !plan.cs31 || !plan.cs11) return MPX_ENOMEM;
    ctx->misc_plans[key] = {(void*)plan.pos, (void*)plan.pairs, (void*)plan.cs31, (void*)plan.cs11};
    return MPX_OK;
}

struct BandCoef;
int remez_taps_for(mpx_ctx* ctx, int fs, double* c13);
static int band_coefs_rest(int fs, BandCoef& k);

// Filter design (host constants): closed forms of dsp/wfir.py:6-21, scipy.signal.butter(2, ...)
static int band_coefs(mpx_ctx* ctx, int fs, BandCoef& k) {
    int rc0 = remez_taps_for(ctx, fs, k.c);
    if (rc0) return rc0;
    return band_coefs_rest(fs, k);
}

// 13 warped-FIR taps = scipy.signal.remez(13, [0,19,20,r,r+1,fs/2], [0,1,0], fs) (dsp/wfir.py:13-21):
// built-in for 22050 / 44100 Hz, otherwise registered by the host through mpx_set_remez_taps.
int remez_taps_for(mpx_ctx* ctx, int fs, double* c13) {
    static const double remez22050[13] = {
        -0.2503141758465685, -0.00010985253437014219, 2.5089273607721457e-05, -0.0002589029075222507,
        0.00020302128904025308, 0.0002957470030712228, 1.0000257474401701, 0.0002957470030712228,
        0.00020302128904025308, -0.0002589029075222507, 2.5089273607721457e-05, -0.00010985253437014219,
        -0.2503141758465685};
    static const double remez44100[13] = {
        -0.28459907723604855, 0.07244737666028533, -0.07937698258360983, 0.0848775348433701,
        -0.08886832893562109, 0.09162147542045991, 0.9077249116355622, 0.09162147542045991,
        -0.08886832893562109, 0.0848775348433701, -0.07937698258360983, 0.07244737666028533,
        -0.28459907723604855};
    const double* taps = fs == 22050 ? remez22050 : (fs == 44100 ? remez44100 : nullptr);
    auto it = ctx->remez.find(fs);
    if (!taps && it == ctx->remez.end())
        return set_error(ctx, MPX_EUNSUPPORTED,
                         "ESACF: no warped-FIR (remez) taps for fs=%d; built-in tables cover 22050 and 44100 Hz "
                         "(register others with mpx_set_remez_taps)", fs);
    for (int i = 0; i < 13; ++i) c13[i] = taps ? taps[i] : it->second[i];
    return MPX_OK;
}

static int band_coefs_rest(int fs, BandCoef& k) {
    k.a = 1.0674 * std::sqrt((2.0 / M_PI) * std::atan(0.06583 * fs / 1000.0)) - 0.1916;
    const double kk = std::tan(M_PI * 1000.0 / fs);
    const double norm = 1.0 / (1.0 + std::sqrt(2.0) * kk + kk * kk);
    const double a1 = 2.0 * (kk * kk - 1.0) * norm, a2 = (1.0 - std::sqrt(2.0) * kk + kk * kk) * norm;
    k.lpb[0] = kk * kk * norm;
    k.lpb[1] = 2.0 * kk * kk * norm;
    k.lpb[2] = kk * kk * norm;
    k.hpb[0] = norm;
    k.hpb[1] = -2.0 * norm;
    k.hpb[2] = norm;
    k.lpa[0] = k.hpa[0] = 1.0;
    k.lpa[1] = k.hpa[1] = a1;
    k.lpa[2] = k.hpa[2] = a2;
    return MPX_OK;
}

// How a frame of N samples is cut in time for bandsplit_kernel (BandCut).  The run-in a later piece needs is read off the
// filters themselves: the impulse responses of input -> residual -> high-pass, input -> residual -> low-pass and of the
// low-pass alone (it follows the rectifier) are simulated, and the run-in is the first multiple of 16 samples t at which
// what a forgotten past can still contribute -- sum_{n >= t} |h[n]|, for the rectified band the low-pass's tail plus the
// high-pass's tail smeared by the low-pass -- is below 2^-60 of sum |h|.  1 kHz Butterworth poles at 44.1 kHz have radius
// 0.904: 512 samples; at 22.05 kHz 0.818: 272.  A frame is cut when it is at least four run-ins long (the reference's own
// 46.4 ms frames -- 1023 / 2046 samples -- are not: their batches are thousands of frames and fill the chip uncut), into
// BS_PIECES pieces of equal WORK: the first one, without run-in, is longer by it.  Nothing here looks at the batch.
constexpr int BS_PIECES = 8;
static BandCut band_cut(mpx_ctx* ctx, int fs, int N, const BandCoef& k) {
    const int ntiles = (N + BS_TILE - 1) / BS_TILE;
    BandCut one{ntiles, 0, 0};
    const int pieces = dev_env_int("MPX_BS_PIECES", BS_PIECES);
    if (pieces <= 1) return one;
    const std::string key = "bs_runin_" + std::to_string(fs);
    auto it = ctx->host_blobs.find(key);
    if (it == ctx->host_blobs.end()) {
        constexpr int T = 16384;
        std::vector<double> hr(T), hhp(T), hlo(T), hlp(T);
        {   // dsp/wfir.py:25-43 on a unit impulse, then the three biquads (esacf.py:47-51)
            double z[12] = {0}, h1 = 0, h2 = 0, l1 = 0, l2 = 0, g1 = 0, g2 = 0;
            for (int t = 0; t < T; ++t) {
                const double x = t == 0 ? 1.0 : 0.0;
                double in = x, xh = k.c[0] * x;
                for (int i = 0; i < 12; ++i) {
                    const double o = -k.a * in + z[i];
                    z[i] = in + k.a * o;
                    xh += k.c[i + 1] * o;
                    in = o;
                }
                const double r = x - xh;
                hr[t] = r;
                const double yh = k.hpb[0] * r + h1;
                h1 = (h2 + k.hpb[1] * r) - k.hpa[1] * yh;
                h2 = k.hpb[2] * r - k.hpa[2] * yh;
                hhp[t] = yh;
                const double yl = k.lpb[0] * r + l1;
                l1 = (l2 + k.lpb[1] * r) - k.lpa[1] * yl;
                l2 = k.lpb[2] * r - k.lpa[2] * yl;
                hlo[t] = yl;
                const double yp = k.lpb[0] * x + g1;
                g1 = (g2 + k.lpb[1] * x) - k.lpa[1] * yp;
                g2 = k.lpb[2] * x - k.lpa[2] * yp;
                hlp[t] = yp;
            }
        }
        auto tails = [&](const std::vector<double>& h) {   // tail[t] = sum_{n >= t} |h[n]| / sum |h|
            std::vector<double> tl(T + 1, 0.0);
            for (int t = T - 1; t >= 0; --t) tl[t] = tl[t + 1] + std::fabs(h[t]);
            const double tot = tl[0] > 0 ? tl[0] : 1.0;
            for (double& v : tl) v /= tot;
            return tl;
        };
        // (thr: the warped-FIR residual itself -- what a cut piece writes for the MPX_STAGE_WFIR tap and what feeds every later
        // stage.  For the built-in tables its all-pass pole decays far faster than the 1 kHz biquads; for taps the caller
        // registered for another rate (mpx_set_remez_taps, which also drops this cache entry) nothing else would bound it.)
        const std::vector<double> thr = tails(hr), thp = tails(hhp), tlo = tails(hlo), tlp = tails(hlp);
        double glp = 0.0;
        for (double v : hlp) glp += std::fabs(v);
        const double eps = std::ldexp(1.0, -60);
        int runin = -1;
        for (int t = BS_TILE; t <= T / 2 && runin < 0; t += BS_TILE) {
            if (thr[t] >= eps || tlo[t] >= eps || tlp[t] >= eps || thp[t] >= eps) continue;
            double smear = 0.0;   // the rectified band: the high-pass's leftover at t - j through tap j of the low-pass
            for (int j = 0; j <= t; ++j) smear += std::fabs(hlp[j]) * thp[t - j];
            if (smear / glp + tlp[t] < eps && tlo[T / 2] < eps * 1e-3) runin = t;
        }
        std::vector<unsigned char> blob(sizeof(int));
        std::memcpy(blob.data(), &runin, sizeof(int));
        it = ctx->host_blobs.emplace(key, std::move(blob)).first;
    }
    int runin;
    std::memcpy(&runin, it->second.data(), sizeof(int));
    if (runin < 0 || N < 4 * runin) return one;
    BandCut c;
    c.rt = runin / BS_TILE;
    c.pt = (ntiles - c.rt + pieces - 1) / pieces;
    c.p0 = ntiles - (pieces - 1) * c.pt;
    if (c.pt < 1 || c.p0 < 1) return one;
    return c;
}
static int band_cut_pieces(const BandCut& c, int N) { return c.pt ? 1 + ((N + BS_TILE - 1) / BS_TILE - c.p0 + c.pt - 1) / c.pt : 1; }

template <int L, bool BLUE>
static int sacf_launch(mpx_ctx* ctx, const SacfArgs& a, long long frames, hipStream_t st) {
    const size_t lds = sizeof(cx<double>) * L + sizeof(double) * (size_t)(a.Mh + 2);
    const size_t alias = peak_scratch_bytes(a.Mh);
    if (alias > sizeof(cx<double>) * L)
        return set_error(ctx, MPX_EUNSUPPORTED, "ESACF: peak-picking scratch does not fit (N=%d)", a.N);
    auto kern = sacf_kernel<L, BLUE>;
    if (lds > 48 * 1024)
        MPX_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)(a.pair ? (frames + 1) / 2 : frames)), dim3(L / 8), lds, st, a);
    MPX_HIP(ctx, hipGetLastError());
    return MPX_OK;
}

// the stream coopfit_live_kernel runs on and its two events exist together or not at all (a failure: the batch runs without it)
static bool side_stream_ready(mpx_ctx* ctx) {
    if (ctx->side_ready) return true;
    bool ok = hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking) == hipSuccess;
    for (int k = 0; ok && k < 2; ++k) ok = hipEventCreateWithFlags(&ctx->side_ev[k], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        for (int k = 0; k < 2; ++k) {
            if (ctx->side_ev[k]) (void)hipEventDestroy(ctx->side_ev[k]);
            ctx->side_ev[k] = nullptr;
        }
        if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
        ctx->side_stream = nullptr;
        return false;
    }
    ctx->side_ready = true;
    return true;
}

int esacf_run(mpx_ctx* ctx, const float* d_signal, int64_t n, const FrameDesc* d_desc, int64_t num_frames,
              int fs, const mpx_esacf_params* params, int frame, int hop, double* d_chroma_frames, int stage,
              double* d_stage_out, hipStream_t st) {
    mpx_esacf_params p = params ? *params : mpx_esacf_params{6, 0.1, 10, MPX_ENHANCE_LIBROSA010, MPX_NOTES_UNICODE};
    // MPX_FLAG_DETERMINISTIC (mpx_create folds MPX_DETERMINISTIC=1 of its environment into it): every gaussian fit is
    // finished on the lane that started it.  Results do not depend on it any more -- the cooperative kernel and the
    // lane kernel round identically (see "ONE arithmetic for both fit kernels") -- it remains as the slower way of
    // computing the same bits, for cross-checks.
    const bool deterministic = (ctx->flags & MPX_FLAG_DETERMINISTIC) != 0;
    const int N = frame, Mh = (N - 1) / 2;
    // above 4096 samples: one radix-2 split around chirp-z transforms of N/2 points (sacf_split_kernel), even N only
    // odd lengths above 4096 and everything above 8192: chirp-z of 16384 / 32768 points around a radix-2 / 4 step (sacf_huge_kernel)
    const bool huge = N > 8192 || (N > 4096 && (N & 1));
    const bool split = N > 4096 && !huge;
    if (N < 64 || N > 16384)
        return set_error(ctx, MPX_EUNSUPPORTED, "ESACF: frame length %d (supported: 64 ... 16384)", N);
    if (p.enhance_mode != MPX_ENHANCE_LIBROSA010 && p.enhance_mode != MPX_ENHANCE_NOOP)
        return set_error(ctx, MPX_EINVAL, "ESACF: unknown enhance_mode %d", p.enhance_mode);
    if (p.peak_min_dist < 0 || p.n_peaks_elim < 0 || p.n_peaks_elim > 64)
        return set_error(ctx, MPX_EINVAL, "ESACF: bad peak parameters");
    if (p.note_names != MPX_NOTES_UNICODE && p.note_names != MPX_NOTES_ASCII)
        return set_error(ctx, MPX_EINVAL, "ESACF: unknown note_names %d", p.note_names);
    if (fs <= 0) return set_error(ctx, MPX_EINVAL, "fs must be positive");
    // librosa.effects.time_stretch is a pure truncation only while its STFT has <= 2 frames; above that
    // (Mh >= 1024 lags) the real phase vocoder runs in its own kernel between the SACF and the peak picking
    const bool pv = p.enhance_mode == MPX_ENHANCE_LIBROSA010 && p.n_peaks_elim >= 2 && 1 + Mh / 512 > 2;
    if (num_frames == 0) return MPX_OK;
    BandCoef coef;
    int rc = band_coefs(ctx, fs, coef);
    if (rc) return rc;
    dev_tick(ctx, "esacf_run: enter");
    const BandCut bcut = band_cut(ctx, fs, N, coef);
    dev_tick(ctx, "esacf_run: band_cut");
    EsacfPlan plan;
    HugeArgs hg{};
    if (huge) {
        constexpr int M = 8192;
        const int R = 2 * N - 1 <= 2 * M ? 2 : 4, L = R * M;
        const std::string key = "esacf_huge" + std::to_string(N);
        auto it = ctx->misc_plans.find(key);
        if (it == ctx->misc_plans.end()) {
            std::vector<cx<double>> tw(M), twL(L), chirp(N), filt(L, cx<double>{0.0, 0.0}), fr(L);
            for (int j = 0; j < M; ++j) {
                const long double ang = -2.0L * M_PIl * j / (long double)M;
                tw[j] = {(double)cosl(ang), (double)sinl(ang)};
            }
            for (int j = 0; j < L; ++j) {
                const long double ang = -2.0L * M_PIl * j / (long double)L;
                twL[j] = {(double)cosl(ang), (double)sinl(ang)};
            }
            for (long long n = 0; n < N; ++n) {
                const long long q = (n * n) % (2LL * N);  // exact phase reduction
                const long double ang = M_PIl * (long double)q / (long double)N;
                chirp[n] = {(double)cosl(ang), (double)sinl(ang)};
            }
            filt[0] = chirp[0];
            for (int m = 1; m < N; ++m) filt[m] = filt[L - m] = chirp[m];
            host_fft(filt);
            for (int r = 0; r < R; ++r)
                for (int k = 0; k < M; ++k) fr[(size_t)r * M + k] = {filt[(size_t)R * k + r].x / L, filt[(size_t)R * k + r].y / L};
            void* d0 = upload(ctx, tw.data(), tw.size() * sizeof(cx<double>));
            void* d1 = upload(ctx, chirp.data(), chirp.size() * sizeof(cx<double>));
            void* d2 = upload(ctx, fr.data(), fr.size() * sizeof(cx<double>));
            void* d3 = upload(ctx, twL.data(), twL.size() * sizeof(cx<double>));
            if (!d0 || !d1 || !d2 || !d3) return MPX_ENOMEM;
            it = ctx->misc_plans.emplace(key, std::vector<void*>{d0, d1, d2, d3}).first;
        }
        plan.tw = (cx<double>*)it->second[0];
        plan.chirp = (cx<double>*)it->second[1];
        plan.bhat = nullptr;
        plan.L = M;
        plan.blue = true;
        hg.bhat_r = (const cx<double>*)it->second[2];
        hg.twL = (const cx<double>*)it->second[3];
        hg.R = R;
        hg.L = L;
    } else if ((rc = esacf_plan(ctx, split ? N / 2 : N, plan, split))) return rc;
    // non-powers of two of 2049 ... 2730 samples: three half-length chirp-z transforms on the in-place 4096-point engine
    const bool rz = !split && !huge && plan.blue && plan.L == 8192 && N + N / 2 <= 4096 && !dev_env_on("MPX_SACF_NO_RZ");
    if (rz) {
        const std::string key = "esacf_rz" + std::to_string(N);
        auto it = ctx->misc_plans.find(key);
        if (it == ctx->misc_plans.end()) {
            constexpr int LR = 4096;
            const int K = N / 2 + 1;
            std::vector<cx<double>> tw(LR), chirp(N), filt(LR, cx<double>{0.0, 0.0}), fr(LR);
            for (int j = 0; j < LR; ++j) {
                const long double ang = -2.0L * M_PIl * j / (long double)LR;
                tw[j] = {(double)cosl(ang), (double)sinl(ang)};
            }
            for (long long n = 0; n < N; ++n) {
                const long long q = (n * n) % (2LL * N);
                const long double ang = M_PIl * (long double)q / (long double)N;
                chirp[n] = {(double)cosl(ang), (double)sinl(ang)};
            }
            for (int m = 0; m < K; ++m) filt[m] = chirp[m];            // chirp[k - n], k - n = 0 .. K-1
            for (int m = 1; m < N; ++m) filt[LR - m] = chirp[m];       // k - n = -1 .. -(N-1): N + K - 1 <= LR points in all
            host_fft(filt);
            for (auto& v : filt) {
                v.x /= LR;
                v.y /= LR;
            }
            for (int t = 0; t < LR / 8; ++t)
                for (int e = 0; e < 8; ++e) fr[(size_t)e * (LR / 8) + t] = filt[dif_reg_freq(LR, t, e)];
            void* d0 = upload(ctx, tw.data(), tw.size() * sizeof(cx<double>));
            void* d1 = upload(ctx, chirp.data(), chirp.size() * sizeof(cx<double>));
            void* d2 = upload(ctx, fr.data(), fr.size() * sizeof(cx<double>));
            if (!d0 || !d1 || !d2) return MPX_ENOMEM;
            it = ctx->misc_plans.emplace(key, std::vector<void*>{d0, d1, d2}).first;
        }
        plan.tw = (cx<double>*)it->second[0];
        plan.chirp = (cx<double>*)it->second[1];
        plan.bhat = (cx<double>*)it->second[2];
    }
    const cx<double>* twn = nullptr;
    if (split) {
        const std::string key = "esacf_twn" + std::to_string(N);
        auto it = ctx->misc_plans.find(key);
        if (it == ctx->misc_plans.end()) {
            std::vector<cx<double>> w((size_t)N / 2 + 1);
            for (int k = 0; k <= N / 2; ++k) {
                const long double ang = -2.0L * M_PIl * k / (long double)N;
                w[k] = {(double)cosl(ang), (double)sinl(ang)};
            }
            void* d = upload(ctx, w.data(), w.size() * sizeof(cx<double>));
            if (!d) return MPX_ENOMEM;
            it = ctx->misc_plans.emplace(key, std::vector<void*>{d}).first;
        }
        twn = (const cx<double>*)it->second[0];
    }
    // the reference's own frame lengths (1023, 2046) run on the prime-factor engine; MPX_SACF_BLUESTEIN=1 forces the chirp-z
    const bool use_pfa = !split && !huge && pfa_supported(N) && !dev_env_on("MPX_SACF_BLUESTEIN");
    PfaPlan pfa;
    if (use_pfa && (rc = pfa_plan(ctx, N, pfa))) return rc;
    if (plan.L > 8192)
        return set_error(ctx, MPX_EUNSUPPORTED, "ESACF: non power-of-two frame %d needs a %d-point FFT (> 8192)", N, plan.L);
    int maxp = p.peak_min_dist > 1 ? Mh / (p.peak_min_dist + 1) + 2 : Mh / 2 + 2;
    if (maxp > 4095 && (Mh - 1) / 2 <= 4095) maxp = 4095;   // peaks sit at 1 .. Mh - 2 and are never adjacent: (Mh - 1) / 2 at most
    if (maxp > 4095) return set_error(ctx, MPX_EUNSUPPORTED, "ESACF: too many peak slots");

    // frames are processed in batches that fit a fixed workspace budget
    const size_t per_frame = (size_t)(N + BS_TILE) * 16 + (size_t)Mh * 8 + (size_t)maxp * 20 + 8;
    long long batch = (long long)((size_t(24) << 30) / per_frame);
    if (batch > num_frames) batch = num_frames;
    if (batch > (1 << 19)) batch = 1 << 19;  // (frame << 12 | slot) must fit an int
    if (batch < 1) batch = 1;
    // (x_lo, x_hi) in band layout: whole blocks of 64 frames x whole tiles of 16 samples
    if ((rc = ensure(ctx, ctx->d_ws0, (size_t)((batch + 63) / 64) * ((N + BS_TILE - 1) / BS_TILE) * 64 * BS_TILE * 16))) return rc;
    if ((rc = ensure(ctx, ctx->d_ws1, (size_t)batch * Mh * 8 + 64))) return rc;              // y
    const long long fit_resident = (long long)ctx->num_cus * (4 * FIT_WAVES_PER_SIMD / (FIT_THREADS / 64));  // blocks
    const size_t park_slots = (size_t)fit_resident * FIT_THREADS + EARLY_PARK_CAP;   // at most one end-game park per lane, and the early ones of a small batch
    const size_t park_bytes = 2 * park_slots * sizeof(ParkedFit);  // a second list for coopfit_kernel's second pass
    // (the work list is WL_SHARDS regions of ceil(batch / WL_SHARDS) * maxp items: up to WL_SHARDS * maxp more than batch * maxp)
    const size_t wl_items = (size_t)WL_SHARDS * (size_t)((batch + WL_SHARDS - 1) / WL_SHARDS) * maxp;
    const size_t shard_bytes = (size_t)WL_SHARDS * WL_SHARD_STRIDE * sizeof(unsigned long long);
    if ((rc = ensure(ctx, ctx->d_ws3, (size_t)batch * maxp * 16 + wl_items * 4 + (size_t)batch * 4 + 512 + shard_bytes + park_bytes))) return rc;
    cx<double>* xb = (cx<double>*)ctx->d_ws0.p;
    double* y = (double*)ctx->d_ws1.p;
    char* w3 = (char*)ctx->d_ws3.p;
    double* center = (double*)w3;
    int* peak_idx = (int*)(w3 + (size_t)batch * maxp * 8);
    int* okf = peak_idx + (size_t)batch * maxp;
    int* worklist = okf + (size_t)batch * maxp;
    int* peak_count = worklist + wl_items;
    int* total = peak_count + batch;  // counters, see SacfArgs::total_peaks; [3] parked fits, [4] next parked fit, [8] fits parked again, [9] next of those
    unsigned long long* shard_cnt = reinterpret_cast<unsigned long long*>(((uintptr_t)(total + 16) + 127) & ~(uintptr_t)127);   // [WL_SHARDS] x 128 B
    ParkedFit* parked = reinterpret_cast<ParkedFit*>(((uintptr_t)(shard_cnt + WL_SHARDS * WL_SHARD_STRIDE) + 63) & ~(uintptr_t)63);
    ParkedFit* parked2 = parked + park_slots;

    for (long long f0 = 0; f0 < num_frames; f0 += batch) {
        const long long nf = (num_frames - f0 < batch) ? num_frames - f0 : batch;
        double* xw = (stage == MPX_STAGE_WFIR) ? d_stage_out + (size_t)f0 * N : nullptr;
        dev_tick(ctx, "esacf_run: before bandsplit");
        prof_mark(ctx, st, "bandsplit_kernel");
        const dim3 bs_grid((unsigned)((nf + 63) / 64), (unsigned)band_cut_pieces(bcut, N));
        if (xw)
            hipLaunchKernelGGL(bandsplit_kernel<true>, bs_grid, dim3(64), 0, st, d_signal,
                               (long long)n, d_desc, f0, nf, N, hop, coef, xb, xw, bcut);
        else
            hipLaunchKernelGGL(bandsplit_kernel<false>, bs_grid, dim3(64), 0, st, d_signal,
                               (long long)n, d_desc, f0, nf, N, hop, coef, xb, xw, bcut);
        MPX_HIP(ctx, hipGetLastError());
        if (stage == MPX_STAGE_XLO || stage == MPX_STAGE_XHI)
            hipLaunchKernelGGL(band_unpack_kernel, dim3((unsigned)nf), dim3(256), 0, st, xb, nf, N, stage == MPX_STAGE_XHI ? 1 : 0,
                               d_stage_out + (size_t)f0 * N);
        if (stage >= 0 && stage <= MPX_STAGE_XHI) continue;
        prof_mark(ctx, st, nullptr);
        MPX_HIP(ctx, hipMemsetAsync(shard_cnt, 0, shard_bytes, st));
        MPX_HIP(ctx, hipMemsetAsync(total, 0, 16 * sizeof(int), st));  // see SacfArgs::total_peaks; [5] parking attempts; [6..7] evaluations; [10] early parks; [11] lane-kernel waves signed off; [12] fits the live kernel finished
        SacfArgs a;
        a.xb = xb;
        a.N = N;
        a.Mh = Mh;
        a.tw = plan.tw;
        a.chirp = plan.chirp;
        a.bhat = plan.bhat;
        a.twn = twn;
        a.n_peaks_elim = p.n_peaks_elim;
        a.peak_thresh = p.peak_thresh;
        a.peak_min_dist = p.peak_min_dist;
        a.enhance_mode = p.enhance_mode;
        a.defer_enhance = pv ? 1 : 0;
        a.maxp = maxp;
        a.sacf_out = stage == MPX_STAGE_SACF ? d_stage_out + (size_t)f0 * Mh : nullptr;
        a.y_out = y;
        a.peak_count = peak_count;
        a.peak_idx = peak_idx;
        a.total_peaks = total;
        a.worklist = worklist;
        a.shard_cap = (int)(((nf + WL_SHARDS - 1) / WL_SHARDS) * maxp);
        a.worklist_cap = WL_SHARDS * a.shard_cap;
        a.shard_cnt = shard_cnt;
        a.num_frames = nf;
        // measured, not adopted: pairing saves 0.85 ms per 176 k frames, but the rounding-level cross-talk between the
        // two frames makes results depend on the batch neighbour and flips 0.075 % of the frames (ill-conditioned fits)
        a.pair = !deterministic && dev_env_on("MPX_SACF_PAIR") ? 1 : 0;
        {
            auto pt = ctx->misc_plans.find("pow067_tab");
            if (pt == ctx->misc_plans.end()) {
                double tab[p067::TAB_DOUBLES];
                p067::build_tables(tab);
                void* d = upload(ctx, tab, sizeof tab);
                if (!d) return MPX_ENOMEM;
                pt = ctx->misc_plans.emplace("pow067_tab", std::vector<void*>{d}).first;
            }
            a.pow_tab = (const double*)pt->second[0];
        }
        a.ablate = dev_env_int("MPX_SACF_ABLATE", 0);
        prof_mark(ctx, st, huge ? "sacf_huge_kernel" : split ? "sacf_split_kernel" : (use_pfa ? "sacf_pfa_kernel" : (rz ? "sacf_rz_kernel" : (plan.L == 8192 ? "sacf_big_kernel" : "sacf_kernel"))));
        if (huge) {
            const size_t lds = sizeof(cx<double>) * lds_slots(8192);
            const long long grid = nf < ctx->num_cus ? nf : ctx->num_cus;   // persistent: one workgroup per CU (139 KB of LDS)
            if ((rc = ensure(ctx, ctx->d_ws4, (size_t)grid * 2 * N * sizeof(cx<double>)))) return rc;
            hg.scratch = (cx<double>*)ctx->d_ws4.p;
            a.pair = 0;
            auto kern = sacf_huge_kernel<512>;
            MPX_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, st, a, hg);
            MPX_HIP(ctx, hipGetLastError());
            const size_t row_lds = sizeof(double) * (size_t)(Mh + 2);
            if (pv) {
                auto pit = ctx->misc_plans.find("pv_tw2048");
                if (pit == ctx->misc_plans.end()) {
                    std::vector<cx<double>> tw(PV_NFFT);
                    for (int j = 0; j < PV_NFFT; ++j) {
                        const long double ang = -2.0L * M_PIl * j / (long double)PV_NFFT;
                        tw[j] = {(double)cosl(ang), (double)sinl(ang)};
                    }
                    void* d = upload(ctx, tw.data(), tw.size() * sizeof(cx<double>));
                    if (!d) return MPX_ENOMEM;
                    pit = ctx->misc_plans.emplace("pv_tw2048", std::vector<void*>{d}).first;
                }
                PvArgs pa;
                pa.y = y;
                pa.Mh = Mh;
                pa.n_peaks_elim = p.n_peaks_elim;
                pa.tw = (const cx<double>*)pit->second[0];
                size_t front = sizeof(cx<double>) * lds_slots(PV_NFFT);
                if (peak_scratch_bytes(Mh) > front) front = peak_scratch_bytes(Mh);
                auto pvk = pv_enhance_big_kernel<8>;
                MPX_HIP(ctx, hipFuncSetAttribute((const void*)pvk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(front + row_lds)));
                prof_mark(ctx, st, "pv_enhance_big_kernel");
                hipLaunchKernelGGL(pvk, dim3((unsigned)nf), dim3(PV_T), front + row_lds, st, pa, a);
            } else {
                auto ek = enhance_pick_big_kernel<512>;
                const size_t e_lds = peak_scratch_bytes(Mh) + row_lds;
                MPX_HIP(ctx, hipFuncSetAttribute((const void*)ek, hipFuncAttributeMaxDynamicSharedMemorySize, (int)e_lds));
                prof_mark(ctx, st, "enhance_pick_big_kernel");
                hipLaunchKernelGGL(ek, dim3((unsigned)nf), dim3(512), e_lds, st, a);
            }
            MPX_HIP(ctx, hipGetLastError());
        } else if (split) {
            const size_t lds = sizeof(cx<double>) * lds_slots(8192);
            if (peak_scratch_bytes(Mh) + sizeof(double) * (size_t)(Mh + 2) > lds)
                return set_error(ctx, MPX_EUNSUPPORTED, "ESACF: peak-picking scratch does not fit (N=%d)", N);
            a.pair = 0;
            auto kern = sacf_split_kernel<8192, 512>;
            MPX_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3((unsigned)nf), dim3(512), lds, st, a);
            MPX_HIP(ctx, hipGetLastError());
        } else if (use_pfa) {
            a.pfa_pos = pfa.pos;
            a.pfa_pairs = pfa.pairs;
            a.pfa_cs31 = pfa.cs31;
            a.pfa_cs11 = pfa.cs11;
            a.pfa_npairs = pfa.npairs;
            a.pair = 0;
            const size_t lds = PFA_LDS_BYTES(N, Mh);
            if (PFA_YV_OFF(Mh) + 8 * (size_t)(Mh + 2) > sizeof(cx<double>) * (size_t)N || Mh > 4 * PFA_T)
                return set_error(ctx, MPX_EUNSUPPORTED, "ESACF: peak-picking scratch does not fit (N=%d)", N);
            const long long g = nf;
            if (pfa.a0 == 2)
                hipLaunchKernelGGL(sacf_pfa_kernel<2>, dim3((unsigned)g), dim3(PFA_T), lds, st, a);
            else
                hipLaunchKernelGGL(sacf_pfa_kernel<1>, dim3((unsigned)g), dim3(PFA_T), lds, st, a);
            MPX_HIP(ctx, hipGetLastError());
        } else if (rz) {
            const size_t lds = sizeof(cx<double>) * 4096 + sizeof(double) * (size_t)(Mh + 2);
            if (peak_scratch_bytes(Mh) > sizeof(cx<double>) * 4096)
                return set_error(ctx, MPX_EUNSUPPORTED, "ESACF: peak-picking scratch does not fit (N=%d)", N);
            a.pair = 0;
            auto kern = sacf_rz_kernel<4096>;
            MPX_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3((unsigned)nf), dim3(512), lds, st, a);
            MPX_HIP(ctx, hipGetLastError());
        } else if (plan.L == 8192) {
            const size_t lds = sizeof(cx<double>) * lds_slots(8192) + sizeof(double) * (size_t)(Mh + 2);
            auto kern = sacf_big_kernel<8192, 512>;
            MPX_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3((unsigned)nf), dim3(512), lds, st, a);
            MPX_HIP(ctx, hipGetLastError());
        } else if (plan.blue) {
            if (plan.L == 512) rc = sacf_launch<512, true>(ctx, a, nf, st);
            else if (plan.L == 1024) rc = sacf_launch<1024, true>(ctx, a, nf, st);
            else if (plan.L == 2048) rc = sacf_launch<2048, true>(ctx, a, nf, st);
            else rc = sacf_launch<4096, true>(ctx, a, nf, st);
        } else {
            if (plan.L == 512) rc = sacf_launch<512, false>(ctx, a, nf, st);
            else if (plan.L == 1024) rc = sacf_launch<1024, false>(ctx, a, nf, st);
            else if (plan.L == 2048) rc = sacf_launch<2048, false>(ctx, a, nf, st);
            else rc = sacf_launch<4096, false>(ctx, a, nf, st);
        }
        if (rc) return rc;
        if (pv && !huge) {
            auto pit = ctx->misc_plans.find("pv_tw2048");
            if (pit == ctx->misc_plans.end()) {
                std::vector<cx<double>> tw(PV_NFFT);
                for (int j = 0; j < PV_NFFT; ++j) {
                    const long double ang = -2.0L * M_PIl * j / (long double)PV_NFFT;
                    tw[j] = {(double)cosl(ang), (double)sinl(ang)};
                }
                void* d = upload(ctx, tw.data(), tw.size() * sizeof(cx<double>));
                if (!d) return MPX_ENOMEM;
                pit = ctx->misc_plans.emplace("pv_tw2048", std::vector<void*>{d}).first;
            }
            PvArgs pa;
            pa.y = y;
            pa.Mh = Mh;
            pa.n_peaks_elim = p.n_peaks_elim;
            pa.tw = (const cx<double>*)pit->second[0];
            const size_t pv_lds = sizeof(cx<double>) * lds_slots(PV_NFFT) + sizeof(double) * (size_t)(Mh + 2);
            static_assert(PV_T == 256, "peak_pick<256> inside pv_enhance_kernel");
            const bool fused = peak_scratch_bytes(Mh) <= sizeof(cx<double>) * lds_slots(PV_NFFT) && !dev_env_on("MPX_PV_SEPARATE_PICK");
            const bool four = (1 + Mh / PV_HOP + 1) / 2 > 2;   // output frames at rate 2 (at most 4: Mh <= 4095)
            auto pvk = four ? (fused ? pv_enhance_kernel<true, 4> : pv_enhance_kernel<false, 4>)
                            : (fused ? pv_enhance_kernel<true, 2> : pv_enhance_kernel<false, 2>);
            MPX_HIP(ctx, hipFuncSetAttribute((const void*)pvk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pv_lds));
            prof_mark(ctx, st, "pv_enhance_kernel");
            hipLaunchKernelGGL(pvk, dim3((unsigned)nf), dim3(PV_T), pv_lds, st, pa, a);
            if (!fused) {
                const size_t pk_lds = peak_scratch_bytes(Mh) + sizeof(double) * (size_t)(Mh + 2);
                prof_mark(ctx, st, "peakpick_kernel");
                hipLaunchKernelGGL(peakpick_kernel<256>, dim3((unsigned)nf), dim3(256), pk_lds, st, a);
            }
            MPX_HIP(ctx, hipGetLastError());
        }
        if (!pv && !huge && (a.ablate & 16)) {
            const size_t pk_lds = peak_scratch_bytes(Mh) + sizeof(double) * (size_t)(Mh + 2);
            hipLaunchKernelGGL(peakpick_kernel<64>, dim3((unsigned)nf), dim3(64), pk_lds, st, a);
        }
        prof_mark(ctx, st, nullptr);
        if (stage == MPX_STAGE_ESACF)
            MPX_HIP(ctx, hipMemcpyAsync(d_stage_out + (size_t)f0 * Mh, y, (size_t)nf * Mh * 8,
                                        hipMemcpyDeviceToDevice, st));
        if (stage >= 0) continue;
        const long long slots = nf * maxp;
        {
            // persistent grid filling every SIMD with FIT_WAVES_PER_SIMD waves; lanes pull peaks until the list is empty
            long long blocks = (slots + FIT_THREADS - 1) / FIT_THREADS;
            if (blocks > fit_resident) blocks = fit_resident;
            if (dev_env_int("MPX_FIT_BLOCKS", 0) > 0 && blocks > dev_env_int("MPX_FIT_BLOCKS", 0)) blocks = dev_env_int("MPX_FIT_BLOCKS", 0);
            // MINPACK: maxfev = 200 (n + 1); the env knobs are for profiling
            const int maxfev = dev_env_int("MPX_FIT_MAXFEV", 200 * (lm::NP + 1));
            const bool park = !deterministic && !dev_env_on("MPX_FIT_NOPARK");
            prof_mark(ctx, st, "peakfit_kernel");
            // Samples in LDS, fvec recomputed: the arrangement of every batch since the end of round 4.  (Round 3 kept the round-2
            // arrangement -- fvec in LDS, the samples re-read from the row: 40 bytes of scratch, 9.7x the compulsory traffic -- for
            // batches below 32 768 frames, where it was 0.1 ms faster; with the cooperative kernels of round 4 it no longer is:
            // 8192 frames 5.54 against 5.51 ms, 2048 frames 3.75 against 3.95.  Development builds keep it behind MPX_FIT_SAMPLES=0.)
#ifdef MPX_DEV_KNOBS
            const bool in_lds = dev_env("MPX_FIT_SAMPLES") ? dev_env_on("MPX_FIT_SAMPLES") : true;
            auto fit_kernel = in_lds ? peakfit_kernel<true> : peakfit_kernel<false>;
#else
            auto fit_kernel = peakfit_kernel<true>;
#endif
            // Round 6, measured and not adopted (coopfit_live_kernel above): fits that look like runaways leave the lane kernel at
            // once (early_nfev), and coopfit_live_kernel takes every parked fit WHILE the lane kernel runs: on a stream of its
            // own, one workgroup of four waves per CU behind the same LDS padding as the cooperative launches below (a wave per
            // SIMD), the lane kernel on half its usual grid with MPX_FIT_LIVE_HALF=1 (ONE wave per SIMD: 256 + 216 registers and
            // 62 + 84 KB of LDS fit side by side whichever of the two the dispatcher places first).
            // (Early parking WITHOUT the live kernel was measured and loses: the lane kernel of the 8192-frame Target is as long
            // with it -- 1.28 against 1.30 ms: its ordinary fits decide that -- and the cooperative launches behind it get twice the
            // fits, 2.12 against 1.54 ms; gpurun_out -> profiles/r6/fit_early_sweep.txt.  The two come together or not at all.)
            // Development builds only (MPX_FIT_LIVE=1, MPX_FIT_LIVE_HALF, MPX_FIT_EARLY_NFEV=20: what was measured), off by default.
            const bool live = park && dev_env_int("MPX_FIT_LIVE", 0) != 0 && side_stream_ready(ctx);
            const int early_nfev = live ? dev_env_int("MPX_FIT_EARLY_NFEV", 0) : 0;
            unsigned live_tag = 0;
#ifdef MPX_DEV_KNOBS
            if (live) {
                if (++ctx->live_epoch == 0) ++ctx->live_epoch;
                live_tag = ctx->live_epoch;
                const long long half = std::max<long long>(1, fit_resident / 2);
                if (blocks > half && dev_env_int("MPX_FIT_LIVE_HALF", 0)) blocks = half;
                const int lp = dev_env_int("MPX_COOP_PAD_KB", COOP_PAD_KB);
                if (lp) MPX_HIP(ctx, hipFuncSetAttribute((const void*)coopfit_live_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lp * 1024));
                MPX_HIP(ctx, hipEventRecord(ctx->side_ev[0], st));                       // the rows and the work list are complete
                MPX_HIP(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->side_ev[0], 0));
                hipLaunchKernelGGL(coopfit_live_kernel, dim3((unsigned)ctx->num_cus), dim3(COOP_THREADS), (size_t)lp * 1024, ctx->side_stream,
                                   parked, total + 3, total + 4, total + 11, (int)(blocks * (FIT_THREADS / 64)), live_tag, y, center, okf,
                                   maxfev, total + 6, dev_env_int("MPX_FIT_LIVE_IDLE_US", 1000),
                                   dev_env("MPX_DEBUG_FITS") ? total + 12 : (int*)nullptr, dev_env_int("MPX_FIT_LIVE_POLL", 7));
            }
#endif
            hipLaunchKernelGGL(fit_kernel, dim3((unsigned)blocks), dim3(FIT_THREADS), 0, st, total, total + 1, worklist,
                               a.shard_cap, (const unsigned long long*)shard_cnt, y, Mh, maxp, peak_idx, center, okf, maxfev, park ? parked : nullptr,
                               total + 3, dev_env_int("MPX_FIT_PARK_NFEV", nf < 2048 ? PARK_NFEV_SMALL : PARK_NFEV),
                               dev_env_int("MPX_FIT_PARK_LIVE", PARK_LIVE),
                               dev_env_int("MPX_FIT_PARK_CAP", PARK_CAP),
                               early_nfev, std::min(dev_env_int("MPX_FIT_EARLY_CAP", EARLY_PARK_CAP), EARLY_PARK_CAP),
                               live ? live_tag : 0u, live ? total + 11 : (int*)nullptr);
            if (live) {   // the cooperative launches below take what the live kernel left (normally nothing): behind BOTH kernels
                MPX_HIP(ctx, hipEventRecord(ctx->side_ev[1], ctx->side_stream));
                MPX_HIP(ctx, hipStreamWaitEvent(st, ctx->side_ev[1], 0));
            }
            if (park) prof_mark(ctx, st, "coopfit_kernel");
            if (park) {  // the runaway fits still open when the list ran dry: eight lanes each (coopfit8_kernel), all at once
                // Lists of up to COOP_SPLIT fits go to coopfit_kernel (16 lanes per fit: four to a wave), longer ones to
                // coopfit8_kernel (eight to a wave).  The host does not know the count: both forms are launched and the one
                // whose range the count is not in returns at once.  Workgroups of four waves, one per CU (LDS padding): a wave
                // per SIMD.  Pass 1: every parked fit, at most COOP_PASS1_TRIPS trips each; pass 2: the fits still open, packed again.
                const int split = dev_env_int("MPX_COOP_SPLIT", COOP_SPLIT);
                const int t16 = dev_env_int("MPX_COOP16_THREADS", COOP_THREADS), c16 = dev_env_int("MPX_COOP16_PER_CU", 1), p16 = dev_env_int("MPX_COOP16_PAD_KB", COOP_PAD_KB);
                const int t8 = dev_env_int("MPX_COOP_THREADS", COOP_THREADS), c8 = dev_env_int("MPX_COOP_PER_CU", 1), p8 = dev_env_int("MPX_COOP_PAD_KB", COOP_PAD_KB);
                if (p16) MPX_HIP(ctx, hipFuncSetAttribute((const void*)coopfit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, p16 * 1024));
                if (p8) MPX_HIP(ctx, hipFuncSetAttribute((const void*)coopfit8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, p8 * 1024));
                const int pass1 = dev_env_int("MPX_COOP_PASS1_TRIPS", COOP_PASS1_TRIPS);
                const int spread2 = dev_env_int("MPX_COOP_PASS2_SPREAD", 0);
                for (int pass = 0; pass < (pass1 > 0 ? 2 : 1); ++pass) {
                    const ParkedFit* src = pass ? parked2 : parked;
                    int* cnt = pass ? total + 8 : total + 3;
                    int* nxt = pass ? total + 9 : total + 4;
                    ParkedFit* again = pass == 0 && pass1 > 0 ? parked2 : (ParkedFit*)nullptr;
                    hipLaunchKernelGGL(coopfit_kernel, dim3((unsigned)(ctx->num_cus * c16)), dim3(t16), (size_t)p16 * 1024, st, src, cnt, nxt, y,
                                       center, okf, maxfev, again, total + 8, pass1, pass ? spread2 : 0, total + 6, 0, split);
                    hipLaunchKernelGGL(coopfit8_kernel, dim3((unsigned)(ctx->num_cus * c8)), dim3(t8), (size_t)p8 * 1024, st, src, cnt, nxt, y,
                                       center, okf, maxfev, again, total + 8, pass1, pass ? spread2 : 0, total + 6, split + 1, 0x7fffffff);
                }
            }
        }
        prof_mark(ctx, st, nullptr);
        if (dev_env("MPX_DEBUG_FITS")) {  // profiling aid: work-list counters of this batch
            int h[13];
            MPX_HIP(ctx, hipMemcpyAsync(h, total, sizeof(h), hipMemcpyDeviceToHost, st));
            MPX_HIP(ctx, hipStreamSynchronize(st));
            fprintf(stderr, "mpx esacf: frames %lld fits %d (queued first: %d) parked %d (early attempts %d), finished by the live kernel %d, "
                            "open after the first cooperative pass %d\n", nf, h[0] + h[2], h[0], h[3], h[10], h[12], h[8]);
        }
        if (ctx->prof_on) {  // fit statistics of the profiled call (mpx_esacf_fit_stats): one small synchronous copy per batch
            unsigned h[8];
            MPX_HIP(ctx, hipMemcpyAsync(h, total, sizeof(h), hipMemcpyDeviceToHost, st));
            MPX_HIP(ctx, hipStreamSynchronize(st));
            ctx->fit_stats[0] += (long long)h[0] + (long long)h[2];
            ctx->fit_stats[1] += (long long)h[6] + ((long long)h[7] << 32);
            ctx->fit_stats[2] += (long long)h[3];
        }
        prof_mark(ctx, st, "scatter_kernel");
        hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)((nf + 63) / 64)), dim3(64), 0, st, f0, nf, fs, Mh, maxp,
                           p.note_names, y, peak_count, peak_idx, center, okf, d_chroma_frames);
        prof_mark(ctx, st, nullptr);
        MPX_HIP(ctx, hipGetLastError());
    }
    return MPX_OK;
}

}  // namespace mpx
