// Iterative-F0 (Klapuri) chroma (reference method 3) in fp64.  Three kernels:
//
//  1. if0_frontend_kernel   one lane per (chunk, channel): the whole per-channel chain of
//                           iterative_f0.py:58-65 sample by sample -- 2x resonator 1, 2x resonator 2
//                           (iterative_f0.py:171-193, swapped-argument quirk A.1 is in the host-built
//                           coefficients), 12-stage warped all-pass + 13-tap FIR residual (dsp/wfir.py),
//                           |.|, (y + butter2-LP(y, fc))/2.  Output [t][channel] so that the 70 lanes of a
//                           chunk store contiguously.  The reference filters the WHOLE signal
//                           sequentially; every stage is stable, so a chunk that starts from zero state W
//                           samples early reproduces the sequential result to fp64 rounding and chunks run in
//                           parallel.  W comes from the slowest pole of the chain the parameters produce
//                           (if0_warmup: the part of the chain's n^3 rho^n response beyond W is <= 1e-13 of the whole;
//                           the defaults give rho = 0.99893 and W = 40960).
//  2. if0_spectrum_kernel   one workgroup per frame: for every channel, Hamming x frame, zero-pad to
//                           2*frame (iterative_f0.py:72-77), real FFT as a frame-point complex LDS FFT,
//                           |X|^power accumulated over channels in registers (iterative_f0.py:80-85).
//  3. if0_periodicity_kernel one workgroup per frame: periodicity.py:48-163 -- interval-halving period
//                           search (one wave per harmonic m for the range maxima), harmonic
//                           cancellation, pitch-class scatter (quirks A.10-A.13, A.18).
#include <chrono>
#include <map>
#include <mutex>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "mpx_fft_dif.hpp"
#include "mpx_if0_tables.hpp"
#include "mpx_internal.hpp"

namespace mpx {

constexpr int IF0_MAXCH = 128;
constexpr long long IF0_CHUNK = 262144;   // largest front-end chunk (samples; multiple of every frame size)
constexpr long long IF0_CHUNK_MIN = 16384;
constexpr long long IF0_WARMUP = 16384;   // SHORTEST zero-state run-in before a chunk that does not start a clip (if0_warmup)
constexpr long long IF0_WARMUP_MAX = 1 << 22;

struct If0ChanCoef {   // per channel, built on the host in double
    double r1b0, r1b2, r1a1, r1a2;   // resonator 1: b = [rho1, 0, -rho1], a = [1, -A cos1, A^2]
    double r2b0, r2a1, r2a2;         // resonator 2: b = [rho2],         a = [1, -A cos2, A^2]
    double lpb0, lpb1, lpb2, lpa1, lpa2;  // butter(2, fc/(fs/2)) low-pass
};

struct If0Chunk {
    long long sig_start;   // first sample of the chunk in the packed signal buffer
    long long clip_start;  // first sample of the clip in the packed signal buffer
    int len;               // samples to produce (<= IF0_CHUNK)
    int clip_left;         // samples of the clip from sig_start on (>= len unless the clip ends inside)
    int warm;              // run-in samples before sig_start (0 at the start of a clip)
    int pad;
    long long yc_row0;     // the chunk's block of the output buffer starts at yc_row0 * channels; inside: [frame][channel][frame_size]:
                           // a frame's channels are one contiguous piece for the spectrum kernel, and a wave of 64 channels writes
                           // inside a 64 x frame_size window (until round 3: [channel][len], rows a whole chunk apart)
};

struct If0Wfir {
    double a;
    double c[13];
};

constexpr int IF0_TW = 64;   // samples per channel the pipelined front end collects in LDS before it stores them

struct If0TailGroup {   // chunks with the same (warm, len) whose leftover channels (channels % 64) share one wave
    int first, count;   // tail_list[first .. first + count)
};

// Time slices (MPX_OPT_IF0_WORKSPACE_BYTES): a launch produces the outputs [t0, t1) of every chunk -- t0, t1 multiples of the
// frame size, counted from the chunk's start -- into a hand-off buffer that holds ONE slice of every chunk (a chunk's block
// starts at yc_row0 = chunk index x slice length), and carries the filter state of every lane from launch to launch in
// `state` ([wave][value][lane] doubles).  state == nullptr: the whole chunk in one launch (t0 = 0, t1 = INT_MAX), the
// layout of If0Chunk.  A slice boundary is a tile boundary (the tile is empty there) and a block boundary of the pipelined
// body; what a lane carries over is exactly what the unsliced loop would have held at that step: same operations on the same
// operands, bit-identical outputs.
struct If0Slice {
    int t0, t1;
    double* state;
};
constexpr int IF0_STATE_PIPE = 66, IF0_STATE_SEQ = 22;   // doubles per lane: pipelined / sequential body

// TAIL = false: a wave of 64 channels of ONE chunk -- chunk, input pointer and output rows are wave-uniform (scalar
// loads, plain pointer arithmetic).  TAIL = true: a wave of leftover channels of several chunks -- per-lane chunk,
// input pointer and an LDS table of output rows.  (One body for both had cost the common case 16 % of its speed.)
template <bool TAIL, bool SLICED, int TW = IF0_TW>   // TW: samples per channel the tile collects (64; 32: half the LDS, see if0_run_host)
__device__ __forceinline__ void if0_frontend_body(const float* __restrict__ sig, const If0Chunk* __restrict__ chunks,
                                                  long long num_chunks, int channels,
                                                  const If0ChanCoef* __restrict__ coefs, const If0Wfir& wf,
                                                  double* __restrict__ yc, const int* __restrict__ tail_list,
                                                  const If0TailGroup* __restrict__ tail_groups,
                                                  double (*tile)[TW + 1], long long* rowbase, long long ck_u, int ch0_u,
                                                  int nch_u, const If0TailGroup g, int lg_nf, const If0Slice sl, const int warm_cap,
                                                  const double* __restrict__ hwin, const int wmask) {
#pragma clang fp contract(off)
    // One wave per (chunk, group of 64 channels), one lane per channel; the channels % 64 left over (6 of the default
    // 70) would fill a wave to 9 %, so the leftovers of up to 64 / (channels % 64) chunks of equal length and run-in
    // share one: lane -> (chunk, channel).  All lanes of a wave share the loop bounds, and the outputs go
    // through an LDS tile [lane][16 samples] so that the buffer can be [chunk][channel][t] -- every channel's
    // samples contiguous for the spectrum kernel -- with 128-byte row segments per store instead of 8-byte ones
    const int lane = threadIdx.x;
    const int nt = channels & 63;
    long long ck;
    int ch, ch0 = 0;
    bool active = true;
    if (!TAIL) {
        ck = ck_u;
        ch0 = ch0_u;
        ch = ch0 + (lane < nch_u ? lane : 0);   // idle lanes shadow channel ch0 (results discarded)
    } else {
        const int sub = lane / nt;
        active = sub < g.count;
        ck = tail_list[g.first + (active ? sub : 0)];   // idle lanes shadow the first chunk (results discarded)
        ch = lane % nt;                                  // the leftover channels are the FIRST channels % 64 (if0_run_host)
    }
    const If0Chunk c = chunks[ck];
    const int c_warm_all = TAIL ? __builtin_amdgcn_readfirstlane(c.warm) : c.warm;
    const int c_warm = c_warm_all < warm_cap ? c_warm_all : warm_cap;   // the leftover channels' own (shorter) run-in
    const int c_len = TAIL ? __builtin_amdgcn_readfirstlane(c.len) : c.len;
    const If0ChanCoef k = coefs[ch];
    // the chunk's block of the output buffer: [frame of the chunk][channel][frame_size] (see If0Chunk)
    if (TAIL) rowbase[lane] = active ? c.yc_row0 * channels + ((long long)ch << lg_nf) : -1;
    double* __restrict__ out = yc + (size_t)c.yc_row0 * channels + ((size_t)ch0 << lg_nf);  // !TAIL: frame 0, channel ch0
    // The chain is 17 filter stages deep (4 resonators, 12 all-passes, rectifier + low-pass).  Evaluated
    // sample by sample it is ONE dependent chain of ~20 fp64 operations per step, and a lone wave pays the
    // full FMA latency on each.  Software pipelining across samples removes that: in iteration tau stage j
    // works on sample tau - j, reading what stage j-1 produced one iteration earlier.  All 17 stage updates
    // of an iteration are then independent of each other (stages are walked in reverse so that an input is
    // consumed before it is overwritten).  Every stage performs exactly the operations of the sequential
    // form on exactly the same operands, so results are bit-identical; only the schedule changes.
    double a1 = 0, a2 = 0, b1 = 0, b2 = 0, c1 = 0, c2 = 0, d1 = 0, d2 = 0, l1 = 0, l2 = 0;
    double z[12], pin[12], pxh[12];   // all-pass state; pipeline registers feeding all-pass i
#pragma unroll
    for (int i = 0; i < 12; ++i) z[i] = pin[i] = pxh[i] = 0.0;
    // the wfir input sample only has to WAIT 13 steps for its residual: a 16-slot ring indexed by the (unrolled,
    // compile-time) step number instead of a 12-register shift chain
    double sob[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) sob[i] = 0.0;
    double qy = 0, qu = 0, qv = 0;             // inputs of resonator stages 1..3
    double fxh = 0;                            // x_hat input of the final stage
    constexpr int DEPTH = 16;                  // output of sample n appears in iteration n + DEPTH
    const float* __restrict__ x = sig + c.sig_start;   // per lane in a wave of leftovers
    constexpr int PF = 16;                     // = the tile width = DEPTH: block tb produces outputs tb-16 .. tb-1
    // the samples of block tb + PF are fetched while block tb runs: a chunk's lanes are often alone on their SIMD
    // (256 two-second clips x 70 channels are 280 waves on 1024 SIMDs), so nothing else hides the load latency
    // (unconditional loads from a clamped position, the zero padding a select: a conditional load is a branch per sample)
    float nx[PF];
    const int x_lim = c.clip_left < c_len ? c.clip_left : c_len;   // samples from x on that exist and are wanted
    auto fetch = [&](int tb) {
        // A lone wave issues one instruction every four clocks whatever its kind, and the clamped address of a sample is eight
        // scalar instructions + two selects: a block that lies inside the clip (all but the last of a chunk) is fetched as 16
        // consecutive floats -- one scalar load on the uniform path -- and only the block across the end takes the per-sample form.
        if (!TAIL && tb + PF <= x_lim) {   // (uniform)
#pragma unroll
            for (int q = 0; q < PF; ++q) nx[q] = x[tb + q];
            return;
        }
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const bool ok = tb + q < x_lim;
            const float v = x[ok ? tb + q : -c_warm];   // x[-warm] is the first sample of the run-in: always inside the clip
            nx[q] = ok ? v : 0.f;
        }
    };
    // Outputs from the end of the clip on are never looked at: the reference pads the FILTERED signal with zeros, and the
    // spectrum kernel replaces every sample from `valid` on by zero before it multiplies.  A wave of one chunk therefore
    // stops at the last real sample (rounded up to a tile) instead of filtering the zero padding up to the whole frame:
    // 44 112 instead of 49 168 steps for a two-second clip at 22.05 kHz.  (A wave of leftover channels of several chunks
    // keeps the common bound.)
    const int c_end_all = TAIL ? c_len : ((x_lim + PF - 1) / PF * PF < c_len ? (x_lim + PF - 1) / PF * PF : c_len);
    const int sl_t0 = SLICED ? sl.t0 : 0;
    if (SLICED && sl_t0 >= c_end_all) return;                        // (uniform) this chunk ended in an earlier slice
    const int c_end = SLICED && sl.t1 < c_end_all ? sl.t1 : c_end_all;   // this launch produces the outputs [sl_t0, c_end)
    // every state variable of the loop, in a fixed order: [wave][value][lane]
    double* __restrict__ st_lane = SLICED ? sl.state + (size_t)blockIdx.x * IF0_STATE_PIPE * 64 + lane : nullptr;
    auto carry = [&](auto&& io) {
        int n = 0;
        io(a1, n++); io(a2, n++); io(b1, n++); io(b2, n++); io(c1, n++); io(c2, n++); io(d1, n++); io(d2, n++);
        io(l1, n++); io(l2, n++); io(qy, n++); io(qu, n++); io(qv, n++); io(fxh, n++);
#pragma unroll
        for (int i = 0; i < 12; ++i) { io(z[i], n++); io(pin[i], n++); io(pxh[i], n++); }
#pragma unroll
        for (int i = 0; i < 16; ++i) io(sob[i], n++);
    };
    int tb_first = -c_warm;
    if (SLICED && sl_t0 > 0) {   // the previous launch ended behind block tb = sl_t0 (outputs up to sl_t0 - 1, inputs up to sl_t0 + 15)
        carry([&](double& v, int n) { v = st_lane[n * 64]; });
        tb_first = sl_t0 + PF;
    }
    fetch(tb_first);
    // the flush: a store instruction writes 64 / TW rows, lane -> (row rsub of the instruction's rows, sample col of the tile)
    constexpr int RPI = 64 / TW;
    const int col = lane & (TW - 1), rsub = lane / TW;
    double wv_next = hwin[col & wmask];   // window of the first tile of this launch (a slice starts on a frame)
    for (int tb = tb_first; tb < c_end + DEPTH; tb += PF) {
        const int t0 = tb - DEPTH;                             // this block produces the outputs t0 .. t0 + 15
        const int tcol = t0 >= 0 ? (t0 & (TW - 1)) : 0;   // their columns in the tile (the run-in's outputs are dropped)
        float xs[PF];
#pragma unroll
        for (int q = 0; q < PF; ++q) xs[q] = nx[q];
        fetch(tb + PF);
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            // ---- final stage (sample tau-16): residual, full-wave rectifier, low-pass, average
            // (every multiply-add below is an EXPLICIT fma and contraction is off in this function: the pipelined, the
            //  sequential and the time-sliced instantiations must round identically, and the compiler's own contraction
            //  differed between them -- 11 of 800 per 16 samples -- once the loop bounds came from a slice)
            {
                double r = sob[(q + 3) & 15] - fxh;   // written 13 steps ago
                r = fabs(r);   // (np.abs; as `r < 0 ? -r : r` it was a compare, a sign flip and a select per sample, fabs is an operand modifier of the next instruction)
                const double lp = fma(k.lpb0, r, l1);
                l1 = fma(-k.lpa1, lp, fma(k.lpb1, r, l2));
                l2 = fma(-k.lpa2, lp, k.lpb2 * r);
                tile[lane][tcol + q] = (r + lp) / 2.0;   // output sample tau - DEPTH = tb - 16 + q
            }
            // ---- all-pass stages 11..0 (samples tau-15 .. tau-4), dsp/wfir.py:25-43
            {
                const double o = fma(-wf.a, pin[11], z[11]);
                z[11] = fma(wf.a, o, pin[11]);
                fxh = fma(wf.c[12], o, pxh[11]);
            }
#pragma unroll
            for (int i = 10; i >= 0; --i) {
                const double o = fma(-wf.a, pin[i], z[i]);
                z[i] = fma(wf.a, o, pin[i]);
                pin[i + 1] = o;
                pxh[i + 1] = fma(wf.c[i + 1], o, pxh[i]);
            }
            // ---- resonator 2 twice, resonator 1 twice (samples tau-3 .. tau), DF2T like scipy.signal.lfilter
            {
                const double sres = fma(k.r2b0, qv, d1);
                d1 = fma(-k.r2a1, sres, d2);
                d2 = -k.r2a2 * sres;
                pin[0] = sres;
                sob[q] = sres;
                pxh[0] = wf.c[0] * sres;
            }
            {
                const double v = fma(k.r2b0, qu, c1);
                c1 = fma(-k.r2a1, v, c2);
                c2 = -k.r2a2 * v;
                qv = v;
            }
            {
                const double u = fma(k.r1b0, qy, b1);
                b1 = fma(-k.r1a1, u, b2);
                b2 = fma(-k.r1a2, u, k.r1b2 * qy);
                qu = u;
            }
            {
                const double xt = (double)xs[q];
                const double y = fma(k.r1b0, xt, a1);
                a1 = fma(-k.r1a1, y, a2);
                a2 = fma(-k.r1a2, y, k.r1b2 * xt);
                qy = y;
            }
        }
        // The tile collects IF0_TW = 64 samples per channel before it goes out: a store instruction then writes 512 contiguous
        // bytes of ONE channel row (64 lanes x 8 B) where 16-sample tiles wrote four 128-byte pieces of four rows -- a thousand
        // waves x 64 rows of scattered 128-byte lines held the whole launch at ~2 TB/s of HBM writes (the front end of the 1 h
        // stream took the same 2.05 TB/s with 64 and with 70 channels).
        if (t0 >= 0 && t0 < c_end && (tcol == TW - PF || t0 + PF >= c_end)) {   // uniform: a full tile, or the chunk's last block
            const int tg = t0 - tcol, ncols = tcol + PF;   // first sample and width of what the tile holds
            const int ts = tg - sl_t0;                      // ... counted from the first output of this launch
            // where sample tg of channel 0 of this frame sits relative to frame 0, channel 0 (a tile never straddles a frame)
            const size_t foff = (((size_t)(ts >> lg_nf) * channels) << lg_nf) + (size_t)(ts & ((1 << lg_nf) - 1));
            // The frame's Hamming window (iterative_f0.py:80) goes onto the samples HERE (round 4): in the flush a lane is a
            // time step, so the 64 rows of the tile share one window value per lane -- one coalesced load per 64 samples and one
            // multiply per store, the same product the spectrum kernel formed (x * w, x = (r + lp) / 2 already rounded) -- and the
            // spectrum kernel no longer loads 64 KB of window per channel.  (As the scale of the tile store in the loop above, the
            // wave-uniform window value became a scalar load per step on the counter the LDS traffic waits on: front end 42 -> 51 ms.)
            // Chirp-z frame sizes keep their window in the spectrum kernel: hwin is a single 1.0 there (wmask = 0).
            // (fetched one tile ahead: under a launch that writes 2 TB/s a load issued here returned after microseconds, and the
            //  first store of the flush waited for it -- 44 -> 55 ms per hour of audio)
            const double wv = wv_next;
            wv_next = hwin[((int)((ts + TW) & ((1 << lg_nf) - 1)) + col) & wmask];
            const size_t lane_off = foff + ((size_t)rsub << lg_nf) + col;   // (TW = 64: foff + lane)
            wave_lds_fence();
#pragma unroll
            for (int h = 0; h < 64 / (16 * RPI); ++h) {
                double tv[16];
                long long rbv[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    tv[q] = tile[(16 * h + q) * RPI + rsub][col];   // row (16 h + q) RPI + rsub, sample tg + col
                    if (TAIL) rbv[q] = rowbase[(16 * h + q) * RPI + rsub];
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int r0 = (16 * h + q) * RPI;   // the instruction's first row (uniform)
                    if (col < ncols) {
                        if (TAIL) {
                            if (rbv[q] >= 0) yc[rbv[q] + foff + col] = tv[q] * wv;
                        } else if (r0 + rsub < nch_u) {
                            (out + ((size_t)r0 << lg_nf))[lane_off] = tv[q] * wv;
                        }
                    }
                }
            }
            wave_lds_fence();
        }
    }
    if (SLICED && c_end < c_end_all) carry([&](double& v, int n) { st_lane[n * 64] = v; });   // more slices of this chunk follow
}

// The same chain in its SEQUENTIAL form -- every sample walks the 17 stages one after the other -- for TWO waves per SIMD
// (<= 256 registers): no pipeline registers between the stages and no 16-slot ring for the residual's input, 88 registers
// fewer than the pipelined body above.  The dependent chain of a sample is ~21 operations deep against 59 fp64 operations
// to issue, the state updates fill its latency, and the second wave fills what is left; the 16 samples of a tile are unrolled,
// so the compiler may still overlap neighbouring samples where registers allow.  Same operations on the same operands.
template <bool TAIL, bool SLICED>
__device__ __forceinline__ void if0_frontend_seq_body(const float* __restrict__ sig, const If0Chunk* __restrict__ chunks,
                                                      int channels, const If0ChanCoef* __restrict__ coefs, const If0Wfir& wf,
                                                      double* __restrict__ yc, const int* __restrict__ tail_list,
                                                      double (*tile)[17], long long* rowbase, long long ck_u, int ch0_u,
                                                      int nch_u, const If0TailGroup g, int lg_nf, const If0Slice sl, const int warm_cap,
                                                      const double* __restrict__ hwin, const int wmask) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x;
    const int nt = channels & 63;
    long long ck;
    int ch, ch0 = 0;
    bool active = true;
    if (!TAIL) {
        ck = ck_u;
        ch0 = ch0_u;
        ch = ch0 + (lane < nch_u ? lane : 0);
    } else {
        const int sub = lane / nt;
        active = sub < g.count;
        ck = tail_list[g.first + (active ? sub : 0)];
        ch = lane % nt;
    }
    const If0Chunk c = chunks[ck];
    const int c_warm_all = TAIL ? __builtin_amdgcn_readfirstlane(c.warm) : c.warm;
    const int c_warm = c_warm_all < warm_cap ? c_warm_all : warm_cap;
    const int c_len = TAIL ? __builtin_amdgcn_readfirstlane(c.len) : c.len;
    const If0ChanCoef k = coefs[ch];
    if (TAIL) rowbase[lane] = active ? c.yc_row0 * channels + ((long long)ch << lg_nf) : -1;
    double* __restrict__ out = yc + (size_t)c.yc_row0 * channels + ((size_t)ch0 << lg_nf);   // frame 0 of the chunk, channel ch0
    double a1 = 0, a2 = 0, b1 = 0, b2 = 0, c1 = 0, c2 = 0, d1 = 0, d2 = 0, l1 = 0, l2 = 0;
    double z[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) z[i] = 0.0;
    const float* __restrict__ x = sig + c.sig_start;
    constexpr int PF = 16, G = 4;   // tile width; samples per trip of the loop (a real loop: the compiler interleaves at most G samples)
    float nx[G];
    const int x_lim = c.clip_left < c_len ? c.clip_left : c_len;
    auto fetch = [&](int t) {
#pragma unroll
        for (int q = 0; q < G; ++q) {
            const bool ok = t + q < x_lim;
            const float v = x[ok ? t + q : -c_warm];
            nx[q] = ok ? v : 0.f;
        }
    };
    // Only whole 16-sample tiles reach the buffer, so the end is rounded UP to a tile: at a chirp-z frame size a clip's last
    // chunk is nfr * NF - t0 samples, which need not be a multiple of 16 (its row is 2^lgp >= the 64-multiple full chunk
    // long, so the rounded tail stays inside the row; at the power-of-two frame sizes c_len is a multiple of 1024).
    const int c_len16 = (c_len + PF - 1) / PF * PF;
    const int c_end_all = TAIL ? c_len16 : ((x_lim + PF - 1) / PF * PF < c_len16 ? (x_lim + PF - 1) / PF * PF : c_len16);
    const int sl_t0 = SLICED ? sl.t0 : 0;
    if (SLICED && sl_t0 >= c_end_all) return;                        // (uniform) this chunk ended in an earlier slice
    const int c_end = SLICED && sl.t1 < c_end_all ? sl.t1 : c_end_all;   // this launch produces the outputs [sl_t0, c_end)
    double* __restrict__ st_lane = SLICED ? sl.state + (size_t)blockIdx.x * IF0_STATE_SEQ * 64 + lane : nullptr;
    auto carry = [&](auto&& io) {
        int n = 0;
        io(a1, n++); io(a2, n++); io(b1, n++); io(b2, n++); io(c1, n++); io(c2, n++); io(d1, n++); io(d2, n++);
        io(l1, n++); io(l2, n++);
#pragma unroll
        for (int i = 0; i < 12; ++i) io(z[i], n++);
    };
    int t_first = -c_warm;
    if (SLICED && sl_t0 > 0) {
        carry([&](double& v, int n) { v = st_lane[n * 64]; });
        t_first = sl_t0;
    }
    const unsigned lane_off = ((unsigned)(lane >> 4) << lg_nf) + (unsigned)(lane & 15);   // < 4 frame sizes: 32 bits
    fetch(t_first);
    double wv_next = hwin[(lane & 15) & wmask];
#pragma unroll 1
    for (int t = t_first; t < c_end; t += G) {
        float xs[G];
#pragma unroll
        for (int q = 0; q < G; ++q) xs[q] = nx[q];
        fetch(t + G);
        const int col = t & (PF - 1);   // the run-in and the chunk start are multiples of 16
#pragma unroll
        for (int q = 0; q < G; ++q) {
            const double xt = (double)xs[q];   // (explicit fmas, contraction off: see the pipelined body)
            const double y = fma(k.r1b0, xt, a1);
            a1 = fma(-k.r1a1, y, a2);
            a2 = fma(-k.r1a2, y, k.r1b2 * xt);
            const double u = fma(k.r1b0, y, b1);
            b1 = fma(-k.r1a1, u, b2);
            b2 = fma(-k.r1a2, u, k.r1b2 * y);
            const double v = fma(k.r2b0, u, c1);
            c1 = fma(-k.r2a1, v, c2);
            c2 = -k.r2a2 * v;
            const double sres = fma(k.r2b0, v, d1);
            d1 = fma(-k.r2a1, sres, d2);
            d2 = -k.r2a2 * sres;
            double in = sres, xh = wf.c[0] * sres;
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const double o = fma(-wf.a, in, z[i]);
                z[i] = fma(wf.a, o, in);
                xh = fma(wf.c[i + 1], o, xh);
                in = o;
            }
            double r = sres - xh;
            r = fabs(r);
            const double lp = fma(k.lpb0, r, l1);
            l1 = fma(-k.lpa1, lp, fma(k.lpb1, r, l2));
            l2 = fma(-k.lpa2, lp, k.lpb2 * r);
            tile[lane][col + q] = (r + lp) / 2.0;
        }
        if (t >= 0 && col == PF - G) {   // uniform; a tile of 16 samples is complete
            const int tb = t - (PF - G) - sl_t0;   // counted from the first output of this launch
            const size_t foff = (((size_t)(tb >> lg_nf) * channels) << lg_nf) + (size_t)(tb & ((1 << lg_nf) - 1));   // as in the pipelined body
            const double wv = wv_next;   // this lane's column of the tile: one window value for its 16 stores, fetched a tile ahead (see the pipelined body)
            wv_next = hwin[((int)((tb + PF) & ((1 << lg_nf) - 1)) + (lane & 15)) & wmask];
            wave_lds_fence();
            constexpr int SG = TAIL ? 2 : 4;   // store instructions per group (their values and row bases are registers)
#pragma unroll
            for (int h = 0; h < PF / SG; ++h) {
                double tv[SG];
                long long rbv[SG];
#pragma unroll
                for (int q = 0; q < SG; ++q) {
                    const int e = (h * SG + q) * 64 + lane, r = e >> 4, cc = e & 15;
                    tv[q] = tile[r][cc];
                    if (TAIL) rbv[q] = rowbase[r];
                }
#pragma unroll
                for (int q = 0; q < SG; ++q) {
                    const int e = (h * SG + q) * 64 + lane, r = e >> 4, cc = e & 15;
                    if (TAIL) {
                        if (rbv[q] >= 0) yc[rbv[q] + foff + cc] = tv[q] * wv;
                    } else if (r < nch_u) {
                        // rows e >> 4 = 4 (SG h + q) + (lane >> 4): a wave-uniform base per store, one per-lane offset for all
                        (out + ((size_t)(4 * (h * SG + q)) << lg_nf) + foff)[lane_off] = tv[q] * wv;
                    }
                }
            }
            wave_lds_fence();
        }
    }
    if (SLICED && c_end < c_end_all) carry([&](double& v, int n) { st_lane[n * 64] = v; });
}

template <bool SLICED>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void if0_frontend2_kernel(
    const float* __restrict__ sig, const If0Chunk* __restrict__ chunks, long long num_chunks, int channels,
    const If0ChanCoef* __restrict__ coefs, If0Wfir wf, double* __restrict__ yc, const int* __restrict__ tail_list,
    const If0TailGroup* __restrict__ tail_groups, int num_tail_groups, int lg_nf, If0Slice sl, int warm_tail,
    const double* __restrict__ hwin, int wmask) {
    __shared__ double tile[64][17];
    __shared__ long long rowbase[64];
    const int full = channels >> 6;
    If0TailGroup g = {0, 0};
    if ((long long)blockIdx.x >= num_tail_groups) {
        const long long b = (long long)blockIdx.x - num_tail_groups;
        if0_frontend_seq_body<false, SLICED>(sig, chunks, channels, coefs, wf, yc, tail_list, tile, rowbase, b / full, (channels & 63) + (int)(b % full) * 64, 64, g, lg_nf, sl, 0x7fffffff, hwin, wmask);
        return;
    }
    g = tail_groups[blockIdx.x];
    if (g.count == 1)
        if0_frontend_seq_body<false, SLICED>(sig, chunks, channels, coefs, wf, yc, tail_list, tile, rowbase, tail_list[g.first], 0,
                                     channels & 63, g, lg_nf, sl, warm_tail, hwin, wmask);
    else
        if0_frontend_seq_body<true, SLICED>(sig, chunks, channels, coefs, wf, yc, tail_list, tile, rowbase, 0, 0, 0, g, lg_nf, sl, warm_tail, hwin, wmask);
}

template <bool SLICED, int TW = IF0_TW>
__global__ __launch_bounds__(64) void if0_frontend_kernel(const float* __restrict__ sig, const If0Chunk* __restrict__ chunks,
                                                          long long num_chunks, int channels,
                                                          const If0ChanCoef* __restrict__ coefs, If0Wfir wf,
                                                          double* __restrict__ yc, const int* __restrict__ tail_list,
                                                          const If0TailGroup* __restrict__ tail_groups, int num_tail_groups, int lg_nf,
                                                          If0Slice sl, int warm_tail, const double* __restrict__ hwin, int wmask,
                                                          int wave_prio) {
    if (wave_prio) __builtin_amdgcn_s_setprio(3);   // next to another kernel's waves (if0_run_host): this wave's chain goes first
    __shared__ double tile[64][TW + 1];
    __shared__ long long rowbase[64];   // TAIL: per lane, index in yc of its output row, -1 for an idle lane
    const int full = channels >> 6;
    If0TailGroup g = {0, 0};
    // The waves of leftover channels come FIRST in the grid: they are the slower ones per step (per-lane streams), and a
    // launch that needs more than one round of waves (1024 two-second clips: 1127 waves on 1024 one-wave SIMDs) should
    // end on the fast kind.  Round 4: the leftover channels are the FIRST channels % 64 of the bank, not the last: quirk A.1
    // makes the resonators' pole radius grow with the channel frequency, so the low channels forget their start within
    // 8192 samples (their low-pass is the slowest pole there) where the top channels need 40 960 -- the leftover waves, 13 %
    // slower per step, no longer carry the longest run-in as well and stopped being the ones a launch waits for.
    if ((long long)blockIdx.x >= num_tail_groups) {
        const long long b = (long long)blockIdx.x - num_tail_groups;
        if0_frontend_body<false, SLICED, TW>(sig, chunks, num_chunks, channels, coefs, wf, yc, tail_list, tail_groups, tile, rowbase,
                                 b / full, (channels & 63) + (int)(b % full) * 64, 64, g, lg_nf, sl, 0x7fffffff, hwin, wmask);
        return;
    }
    g = tail_groups[blockIdx.x];
    if (g.count == 1)   // a lone set of leftover channels (small batches: the host does not pack them) on the uniform path
        if0_frontend_body<false, SLICED, TW>(sig, chunks, num_chunks, channels, coefs, wf, yc, tail_list, tail_groups, tile, rowbase,
                                 tail_list[g.first], 0, channels & 63, g, lg_nf, sl, warm_tail, hwin, wmask);
    else
        if0_frontend_body<true, SLICED, TW>(sig, chunks, num_chunks, channels, coefs, wf, yc, tail_list, tail_groups, tile, rowbase, 0, 0,
                                0, g, lg_nf, sl, warm_tail, hwin, wmask);
}

// ------------------------------------------------------------------ spectrum
struct If0Frame {
    long long yc_base;  // index in yc of the frame's first sample of channel 0
    int valid;          // samples of the frame that exist (tail of a clip is zero padded)
    int clip;
    int ch_stride;      // distance between channels (= the frame size: [frame][channel][frame_size])
    int pad;
};

#ifdef MPX_DEV_KNOBS
template <int NF, int T>   // NF = frame size = complex FFT length (2*NF real points, upper half zero)
__global__ __launch_bounds__(T) void if0_spectrum_kernel(const double* __restrict__ yc, const If0Frame* __restrict__ frames,
                                                         int channels, double power, const double* __restrict__ window,
                                                         const cx<double>* __restrict__ tw,   // W_NF
                                                         const cx<double>* __restrict__ twn,  // W_{2NF}^k, k <= NF
                                                         double* __restrict__ ut) {           // [F, 2*NF]
    constexpr int M = NF;           // complex points
    constexpr int EPT = M / T;
    constexpr int KPT = (M + T) / T;  // bins k = tid + j*T, k <= M  (M/T + 1 slots)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cx<double>* buf = reinterpret_cast<cx<double>*>(smem);
    const int tid = threadIdx.x;
    const If0Frame fr = frames[blockIdx.x];
    double acc[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) acc[j] = 0.0;
    for (int ch = 0; ch < channels; ++ch) {
        cx<double> regs[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int p = first_pass_index<M, T>(tid, e);  // complex point p = real samples 2p, 2p+1
            const int s = 2 * p;
            double x0 = 0.0, x1 = 0.0;
            if (s < NF) {  // the second half of the 2*NF-point frame is the zero padding
                if (s < fr.valid) x0 = yc[fr.yc_base + (size_t)ch * fr.ch_stride + s];       // (windowed by the front end)
                if (s + 1 < fr.valid) x1 = yc[fr.yc_base + (size_t)ch * fr.ch_stride + s + 1];
            }
            regs[e] = {x0, x1};
        }
        fft_lds<M, T, true, double>(buf, tw, regs, tid);
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const int k = tid + j * T;
            if (k <= M) {
                const cx<double> A = buf[lds_slot(k & (M - 1))];
                cx<double> B = buf[lds_slot((M - k) & (M - 1))];
                B.y = -B.y;
                const cx<double> E = {0.5 * (A.x + B.x), 0.5 * (A.y + B.y)};
                const cx<double> D = {0.5 * (A.x - B.x), 0.5 * (A.y - B.y)};
                const cx<double> X = cadd(E, mul_mi(cmul(twn[k], D)));
                const double mag = hypot(X.x, X.y);
                acc[j] += power == 1.0 ? mag : pow(mag, power);
            }
        }
        __syncthreads();  // buf is rewritten by the next channel's first pass
    }
    double* row = ut + (size_t)blockIdx.x * 2 * NF;
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const int k = tid + j * T;
        if (k <= M) {
            row[k] = acc[j];
            if (k > 0 && k < M) row[2 * NF - k] = acc[j];  // |X[N-k]| = |X[k]| for a real frame
        }
    }
}

#endif  // MPX_DEV_KNOBS

// ------------------------------------------------------------------ periodicity
struct If0PerArgs {
    const double* ut;   // [F, n]
    double* ur;         // [num_slots, n] residual spectrum (scratch of a workgroup)
    double* ud;         // [num_slots, n] detected spectrum (scratch of a workgroup)
    int n;              // 2*frame_size
    double fs, K;       // K = window_size / fs
    double wsize;       // window_size (frame_size) as a double
    int max_voices, Q, M;
    int note_names;  // MPX_NOTES_*
    double tau_min, tau_max, tau_prec, epsilon1, epsilon2, gamma;
    double* chroma;     // [F, 12]
    const int* out_row; // frame f of this launch is row out_row[f] of chroma (nullptr: row f)
    double* voices = nullptr;   // [F, 16] saliences (8) and periods (8) of the detected voices (periodicity.py:112), or nullptr
    long long num_frames;
    unsigned* slot_busy;   // [num_slots] 0 = free (all free between launches)
    int num_slots;
};

constexpr int PER_T = 256;
__constant__ double IF0_HAMMING9[9] = {0.0011244659258033, 0.11559343551383, 0.42817348241183, 0.81822361914331, 1.0,
                                       0.81822361914331, 0.42817348241183, 0.11559343551383, 0.0011244659258033};

// maximum over the wave of values that are never NaN, in every lane: four DPP row steps and four v_readlane -- no LDS
// round trips (a shuffle-based reduction of a double is twelve ds_bpermute with a wait each)
template <int CTRL>
__device__ __forceinline__ double if0_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double if0_readlane(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double if0_wave_max(double v) {
    v = fmax(v, if0_dpp<0xB1>(v));    // quad_perm [1,0,3,2]
    v = fmax(v, if0_dpp<0x4E>(v));    // quad_perm [2,3,0,1]
    v = fmax(v, if0_dpp<0x141>(v));   // row_half_mirror
    v = fmax(v, if0_dpp<0x140>(v));   // row_mirror: all 16 lanes of a row agree
    return fmax(fmax(if0_readlane(v, 0), if0_readlane(v, 16)), fmax(if0_readlane(v, 32), if0_readlane(v, 48)));
}

// One workgroup per frame.  Until the end of round 3 the kernel was bound by VALU ISSUE (3.8 G wave instructions per 6144
// frames: 85 % of its time at four waves per SIMD), and 40 % of them were every thread of the workgroup repeating the same
// serial work after each interval halving: the salience sums -- 2 (M - 1) divisions, one after the other -- and the arg-max over the intervals.  Here wave 0
// does that alone: the weights m fs / tau_up + eps2 of both intervals in ONE parallel step (lane m), the two sums by lane 0
// as the same fused multiply-adds in the same order, the interval bookkeeping and the arg-max with them; the other waves
// wait at the barrier (two per halving step instead of four).  The bounds of the 2 (M - 1) bin ranges are computed one pair
// per lane (two divisions in all instead of two per range) and handed out by v_readlane.  Same operations on the same
// operands: bit-identical chroma (checked against that kernel on a 600 s stream, 1024 clips and two other parameter sets:
// scripts/dev/if0_ab.py); 4.75 -> 3.8 ms per 3230 frames, 8.2 -> 6.6 ms per 6144.
// maxima of a wave's 64 consecutive bins (lane = bin i0 + lane, -inf past nl): the eight 8-bin blocks into e8, the 64-bin block into b64
__device__ __forceinline__ void if0_block_maxima(double m, int lane, int i0, int nl, double* e8, double* b64) {
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
        const double o = __shfl_xor(m, off);
        m = o > m ? o : m;
    }
    if ((lane & 7) == 0 && i0 + lane < nl) e8[(i0 + lane) >> 3] = m;
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
        const double o = __shfl_xor(m, off);
        m = o > m ? o : m;
    }
    if (lane == 0 && i0 + 64 <= nl) b64[i0 >> 6] = m;
}

// BIG (round 6): spectra of more than 16 384 bins (frame sizes 8193 ... 16384): twice the tables, one more level of the sparse
// table; the default instantiation is the round-5 kernel unchanged.
#ifndef IF0_PER_WGS
#define IF0_PER_WGS 4   // workgroups per CU the register allocation is held to (A/B: scripts/dev/spill_ab.sh builds a library with 3 = scratch-free)
#endif
template <bool BIG>
__global__ __launch_bounds__(PER_T, BIG ? 2 : IF0_PER_WGS) void if0_periodicity_kernel(If0PerArgs a) {   // (four workgroups per CU = 128 registers: the gather of the overlapping-window case had taken the kernel to 129 and three -- 2.61 -> 3.05 ms per 600 s; a register cap brings 2.70 back, the gather as a function of its own the same)
    __shared__ double tau_low[32], tau_up[32], smax[32];
    __shared__ double um[128];    // [interval * 64 + harmonic]: range maxima
    __shared__ double wts[128];   // [interval * 64 + harmonic]: m fs / tau_up + epsilon2
    // Range maxima in O(1) loads per range (round 5): e8 = maxima of the 8-bin blocks of the residual, st[k][b] = maximum of the 64-bin
    // blocks b .. b + 2^k - 1 (a sparse table; st[0] is what bmax was).  A range [lo, hi] is at most 7 + 7 bins, 7 + 7 entries of e8
    // and two entries of st -- few enough for ONE LANE per range (see lane_range_max).
    constexpr int NB = BIG ? 512 : 256, LEV = BIG ? 9 : 8;   // 64-bin blocks of a row, levels of the sparse table
    __shared__ double e8[NB * 8];      // n <= 16384 (BIG: 32768)
    __shared__ double st[LEV][NB];
    __shared__ int qbest_sh;
    __shared__ unsigned char dirty[NB];   // 64-bin blocks of the residual a cancellation step has changed
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = a.n;
    // One workgroup per frame, as before -- but the pair of scratch rows (residual and detected spectrum, 2 x 128 KB at the
    // default frame size) belongs to a SLOT the workgroup takes at its start and gives back at its end, not to the frame:
    // 2 x (resident workgroups) slots, 0.5 GB, where every frame had its own pair (5 GB per hour of audio: with the
    // hand-off buffer bounded by time slices this was the next largest workspace, and it went through HBM once per frame).
    // (A persistent grid with one pair per workgroup measured SLOWER -- 23.8-24.8 against 19.8 ms per hour of audio, with
    //  static shares and with an atomic frame counter alike -- so the frames stay the hardware's to schedule.)
    __shared__ int slot_sh;
    const long long f = blockIdx.x;
    if (tid == 0) {
        unsigned s0 = (unsigned)(blockIdx.x * 2654435761u) % (unsigned)a.num_slots;
        for (;;) {   // a free slot always exists: there are twice as many as workgroups can be resident
            if (atomicCAS(&a.slot_busy[s0], 0u, 1u) == 0u) break;
            s0 = s0 + 1 == (unsigned)a.num_slots ? 0u : s0 + 1;
        }
        slot_sh = (int)s0;
    }
    __syncthreads();
    const int slot = slot_sh;
    // What the search ever READS of the residual are the bins up to (M - 1) K / tau_min (the upper end of harmonic M - 1's range
    // in the interval that starts at tau_min: 7411 of the 16 384 at the defaults), but the cancellation walked the partials of a
    // voice up to the top of the spectrum, and the rows were initialised over all of it.  A window changes bins within four of
    // its centre and takes its amplitude from its centre bin, so what happens above a bound can reach down by four bins per
    // voice (<= 8 voices): rows are initialised, partials cancelled and block maxima kept below nl = that bin + 128 only --
    // everything a decision reads is computed exactly as before, 45 % of the row traffic and of the partials at the defaults.
    const int hi_read = (int)((a.M - 1) * a.K / a.tau_min + 0.5);
    const int nl = hi_read + 128 + 63 < n ? ((hi_read + 128 + 63) & ~63) : n;
    double* __restrict__ ur = a.ur + (long long)slot * n;
    double* __restrict__ ud = a.ud + (long long)slot * n;
    {
    const double* __restrict__ uk = a.ut + f * (long long)n;
    // residual = spectrum, detected = 0, and the block maxima of the residual in the same pass (a wave copies whole 64-bin
    // blocks, so it holds each block's maximum: until round 4 build_bmax read the row back) -- for the bins below nl only
    // (four blocks per pass: the stamps had this loop at 80 k clocks per frame -- thirty rounds of one load and its six dependent
    //  exchange steps per wave)
    for (int i0 = wave * 64; i0 < nl; i0 += 4 * PER_T) {
        double m[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * PER_T + lane;
            m[u] = i < nl ? uk[i] : -INFINITY;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * PER_T + lane;
            if (i < nl) {
                ur[i] = m[u];
                ud[i] = 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * PER_T < nl) if0_block_maxima(m[u], lane, i0 + u * PER_T, nl, e8, st[0]);   // (uniform)
    }
    if (tid < 256) dirty[tid] = 0;
    if constexpr (BIG) dirty[256 + tid] = 0;
    __syncthreads();
    __threadfence_block();
    const int nb64 = nl >> 6;   // whole 64-bin blocks (what a range's block part can reach: hi < nl)
    auto build_levels = [&]() {   // st[k] from st[k - 1]; one barrier per level (eight per voice at most)
        for (int k = 1; k < LEV; ++k) {
            const int half = 1 << (k - 1);
            if (tid + 2 * half <= nb64) {
                const double x = st[k - 1][tid], y = st[k - 1][tid + half];
                st[k][tid] = y > x ? y : x;
            }
            if constexpr (BIG) {
                const int t2 = tid + PER_T;
                if (t2 + 2 * half <= nb64) {
                    const double x = st[k - 1][t2], y = st[k - 1][t2 + half];
                    st[k][t2] = y > x ? y : x;
                }
            }
            __syncthreads();
        }
    };
    build_levels();

    // periodicity.py:144-163, the range maxima of TWO intervals (a halving step of min_search evaluates the new interval and the
    // best one so far): 2 (M - 1) ranges, ONE LANE each (pair p = 2 (m - 1) + interval in lane p of wave 0, and p + 64 when M > 33).
    // Until round 5 a wave took a range at a time -- 64 lanes on its ragged ends and block maxima, a wave-wide maximum, ten ranges
    // per wave and step: 17-19 k of a step's 24 k clocks (s_memtime stamps) at four resident workgroups per CU.  A maximum does
    // not depend on the order it is taken in: the same bits.  (Loads unconditional: clamped positions, -inf by select.)
    auto lane_range_max = [&](int lo, int hi) -> double {
        double mm = -INFINITY;
        const int a8 = (lo + 7) >> 3, z8 = (hi + 1) >> 3;
        const bool whole8 = z8 > a8;
        // bins [lo, le) and [rs, hi]: the ragged ends of the whole 8-bin blocks [a8, z8) -- or, without a whole block, the range
        // itself (at most 14 bins) in two sevens.  Fourteen loads in flight, one wait.
        const int le = whole8 ? 8 * a8 : hi + 1, rs = whole8 ? 8 * z8 : lo + 7;
        {
            double v[14];
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                // (32-bit byte offsets from the row's uniform base: one address register per load)
                v[j] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(ur) + (unsigned)((lo + j < le ? lo + j : 0) << 3));
                v[7 + j] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(ur) + (unsigned)((rs + j <= hi ? rs + j : 0) << 3));
            }
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                mm = lo + j < le && v[j] > mm ? v[j] : mm;
                mm = rs + j <= hi && v[7 + j] > mm ? v[7 + j] : mm;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const int a64 = (a8 + 7) >> 3, z64 = z8 >> 3;
        const bool whole64 = whole8 && z64 > a64;
        // 8-bin blocks [a8, le8) and [rs8, z8), likewise
        const int le8 = whole64 ? 8 * a64 : z8, rs8 = whole64 ? 8 * z64 : a8 + 7;
        {
            double v[14];
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                v[j] = e8[whole8 && a8 + j < le8 ? a8 + j : 0];
                v[7 + j] = e8[whole8 && rs8 + j < z8 ? rs8 + j : 0];
            }
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                mm = whole8 && a8 + j < le8 && v[j] > mm ? v[j] : mm;
                mm = whole8 && rs8 + j < z8 && v[7 + j] > mm ? v[7 + j] : mm;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        {
            const int cnt = whole64 ? z64 - a64 : 1;
            const int kk = 31 - __builtin_clz(cnt);
            const int k = kk < LEV - 1 ? kk : LEV - 1;   // (256 blocks -- n = 16384, a range over all of them -- are two windows of 128)
            const double v1 = st[k][whole64 ? a64 : 0], v2 = st[k][whole64 ? z64 - (1 << k) : 0];
            mm = whole64 && v1 > mm ? v1 : mm;
            mm = whole64 && v2 > mm ? v2 : mm;
        }
        return mm;
    };
    auto range_maxima = [&](double tl0, double tl1, double tu0, double tu1) {   // wave 0
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 0 || 2 * (a.M - 1) > 64) {
                const int pidx = lane + 64 * h, m = 1 + (pidx >> 1);
                const double tlw = (pidx & 1) ? tl1 : tl0, tuw = (pidx & 1) ? tu1 : tu0;
                const double tau = 0.5 * (tlw + tuw), deltatau = tuw - tlw;
                const int lo = (int)(m * a.K / (tau + 0.5 * deltatau) + 0.5);
                int hi = (int)(m * a.K / (tau - 0.5 * deltatau) + 0.5);
                if (hi > n - 1) hi = n - 1;  // numpy slicing clips silently
                const bool live = pidx < 2 * (a.M - 1);
                const double mm = lane_range_max(live ? lo : 0, live ? hi : 0);
                if (live) um[(pidx & 1) * 64 + m] = mm;
            }
        }
    };

    __shared__ double voice_sal[8], voice_per[8];   // (LDS: in every thread's registers they cost the search its loads in flight)
    if (tid < 8) voice_sal[tid] = voice_per[tid] = 0.0;
    int voices = 0;
    double prevmix = 0.0, mix = 0.0;
    for (;;) {
        // ---- min_search (periodicity.py:114-142)
        if (tid == 0) {
            tau_low[0] = a.tau_min;
            tau_up[0] = a.tau_max;
            qbest_sh = 0;
        }
        __syncthreads();
        // wave 0 runs the whole search alone, without a workgroup barrier: it holds every range of a step, and the serial part was its already
        if (wave == 0) {
        int q = 0;
        for (;;) {
            const int qbest = qbest_sh;
            const double tlb = tau_low[qbest], tub = tau_up[qbest];
            if (!((tub - tlb) > a.tau_prec && q < a.Q - 1)) break;
            ++q;
            // the new interval q is the upper half of the best one, which keeps its lower half
            const double tlq = (tlb + tub) * 0.5;
            range_maxima(tlq, tlb, tub, tlq);
            wave_lds_fence();
            {
                if (lane >= 1 && lane < a.M) {
                    wts[lane] = lane * a.fs / tub + a.epsilon2;        // interval q:     tau_up = tub
                    wts[64 + lane] = lane * a.fs / tlq + a.epsilon2;   // interval qbest: tau_up = tlq
                }
                wave_lds_fence();
                if (lane == 0) {
                    double s0 = 0.0, s1 = 0.0;
#pragma unroll 4
                    for (int m = 1; m < a.M; ++m) {   // (unrolled: the LDS reads of four harmonics in flight, the sums in order)
                        s0 = __builtin_fma(wts[m], um[m], s0);
                        s1 = __builtin_fma(wts[64 + m], um[64 + m], s1);
                    }
                    tau_low[q] = tlq;
                    tau_up[q] = tub;
                    tau_up[qbest] = tlq;
                    smax[q] = s0 * (a.fs / tlq + a.epsilon1);
                    smax[qbest] = s1 * (a.fs / tlb + a.epsilon1);
                    int whichq = 0;
                    double maxval = smax[0];
#pragma unroll 4
                    for (int j = 1; j <= q; ++j) {
                        const double valnow = smax[j];
                        if (valnow > maxval) {
                            maxval = valnow;
                            whichq = j;
                        }
                    }
                    qbest_sh = whichq;
                }
            }
            wave_lds_fence();
        }
        }
        __syncthreads();
        const int qbest = qbest_sh;
        const double tau = (tau_low[qbest] + tau_up[qbest]) * 0.5;
        const double best = smax[qbest];
        __syncthreads();
        if (voices < 8 && tid == 0) {
            voice_sal[voices] = best;
            voice_per[voices] = tau;
        }
        ++voices;
        mix += best;
        const double test = mix / pow((double)voices, a.gamma);
        if (voices >= a.max_voices || test <= prevmix) break;
        prevmix = test;
        // ---- harmonic cancellation (periodicity.py:76-99); partials of different m never overlap
        const int topm = (int)(tau * (a.fs / a.wsize) * n);  // periodicity.py:78
        const double srovertau = a.fs / tau;
        const double weight = srovertau + a.epsilon1;
        // The 9-bin windows of neighbouring partials are K / tau bins apart: below nine bins (f0 under ~48 Hz at the default
        // frame size; ordinary bass notes at small chirp-z frames, where the spacing can fall below ONE bin) they overlap, and
        // threads scattering their partials would race on the shared bins -- across waves, between the lanes of a wave
        // (loads hoisted over another lane's store), and inside one store instruction when two lanes hit the same bin.  The
        // overlapping case is therefore a GATHER: a thread owns bins, and adds the contributions of the partials that cover a
        // bin in ascending m -- the order of the reference's sequential loop (periodicity.py:83-96), so the sums round alike.
        const bool overlap = a.K / tau < 9.0;   // uniform
        // partials whose window lies below nl: m K / tau + 0.5 < nl - 4 (checked exactly per partial below)
        const int mlim = (int)((nl - 4.5) * tau / a.K) + 2;
        const int mend = nl < n && mlim < topm ? mlim : topm;
        if (!overlap) {
            for (int m = 1 + tid; m < mend; m += PER_T) {
                const double partialK = m * a.K / tau + 0.5;
                if (partialK <= n && partialK < nl - 4) {
                    const int ip = (int)partialK;
                    if (ip < n) {  // the reference would raise IndexError at exactly n; unreachable with the defaults
                        double urw = ur[ip];
                        urw *= weight / (m * srovertau + a.epsilon2);
                        int lowk = (int)(partialK - 4);
                        if (lowk < 0) lowk = 0;
                        int highk = (int)(partialK + 4);
                        if (highk > n) highk = n;
                        for (int j = lowk; j <= highk && j < n; ++j) ud[j] += IF0_HAMMING9[(int)(j - partialK + 4)] * urw;
                    }
                }
            }
        } else {
            const double spacing = a.K / tau;   // bins between neighbouring partials
            const int jend = nl < n ? nl : n;   // (a window below nl ends below nl)
            for (int j = tid; j < jend; j += PER_T) {
                // partials whose window [int(pK - 4), int(pK + 4)] can hold bin j have pK in (j - 5, j + 5): a superset of m,
                // every candidate checked with the reference's own integer arithmetic
                int m0 = (int)floor((j - 5.5) / spacing), m1 = (int)ceil((j + 4.5) / spacing);
                if (m0 < 1) m0 = 1;
                if (m1 > mend - 1) m1 = mend - 1;
                double acc = ud[j];
                for (int m = m0; m <= m1; ++m) {
                    const double partialK = m * a.K / tau + 0.5;
                    if (!(partialK <= n && partialK < nl - 4)) continue;
                    const int ip = (int)partialK;
                    if (ip >= n) continue;
                    int lowk = (int)(partialK - 4);
                    if (lowk < 0) lowk = 0;
                    int highk = (int)(partialK + 4);
                    if (highk > n) highk = n;
                    if (j < lowk || j > highk) continue;
                    double urw = ur[ip];
                    urw *= weight / (m * srovertau + a.epsilon2);
                    acc += IF0_HAMMING9[(int)(j - partialK + 4)] * urw;
                }
                ud[j] = acc;
            }
        }
        __syncthreads();
        __threadfence_block();
        // residual = max(spectrum - detected, 0) changes only where `detected` just did: the same windows again (a bin two
        // windows share is written twice with one value), and the block maxima of the blocks they touch.  Until round 4 this
        // was a pass over the whole row (read two rows, write one) and a second one for the maxima -- per voice; the rows
        // live behind the L2 (a CU's four workgroups hold 1.5 MB of them), and the kernel waited 76 % of its cycles.
        for (int m = 1 + tid; m < mend; m += PER_T) {
            const double partialK = m * a.K / tau + 0.5;
            if (partialK <= n && (int)partialK < n && partialK < nl - 4) {
                int lowk = (int)(partialK - 4);
                if (lowk < 0) lowk = 0;
                int highk = (int)(partialK + 4);
                if (highk > n) highk = n;
                for (int j = lowk; j <= highk && j < n; ++j) {
                    const double d = uk[j] - 1.0 * ud[j];  // cancellation_weight = 1.0
                    ur[j] = d > 0.0 ? d : 0.0;
                    dirty[j >> 6] = 1;   // (index <= NB - 1: n <= 64 NB)
                }
            }
        }
        __syncthreads();
        __threadfence_block();
        for (int b = wave; b < ((nl + 63) >> 6); b += PER_T / 64) {
            if (!dirty[b]) continue;   // uniform over the wave
            const int i = b * 64 + lane;
            if0_block_maxima(i < nl ? ur[i] : -INFINITY, lane, b * 64, nl, e8, st[0]);
            if (lane == 0) dirty[b] = 0;
        }
        __syncthreads();
        build_levels();
        __syncthreads();
    }
    if (tid == 0) {
        double chroma[12];
        for (int i = 0; i < 12; ++i) chroma[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i < a.max_voices) {
                const double pitch = a.fs / voice_per[i];  // quirk A.10: tau is in seconds
                const double midi = 12.0 * (log2(pitch) - log2(440.0)) + 69.0;
                if (midi == midi && !isinf(midi)) {  // unused voices: fs/0 = inf -> OverflowError -> skipped (A.13)
                    const long long note = (long long)nearbyint(midi);
                    const int pc = (int)(((note % 12) + 12) % 12);
                    // A.18: unicode note names drop the sharps
                    if (a.note_names == MPX_NOTES_ASCII || !(pc == 1 || pc == 3 || pc == 6 || pc == 8 || pc == 10))
                        chroma[pc] += voice_sal[i];
                }
            }
        }
        for (int i = 0; i < 12; ++i) a.chroma[(a.out_row ? (long long)a.out_row[f] : f) * 12 + i] = chroma[i];
        if (a.voices) {   // what IterativeF0PeriodicityAnalysis.compute returns next to the chromagram (periodicity.py:112)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                a.voices[(a.out_row ? (long long)a.out_row[f] : f) * 16 + i] = voice_sal[i];
                a.voices[(a.out_row ? (long long)a.out_row[f] : f) * 16 + 8 + i] = voice_per[i];
            }
        }
    }
    }
    __syncthreads();   // everybody's last access to the slot's rows
    if (tid == 0) {
        __threadfence();
        atomicExch(&a.slot_busy[slot], 0u);
    }
}

// ------------------------------------------------------------------ host side
struct If0Plan {
    If0ChanCoef* d_coefs = nullptr;
    double* d_window = nullptr;
    double* d_one = nullptr;         // a single 1.0: the front end's "window" for the chirp-z frame sizes (they window in the spectrum kernel)
    cx<double>* d_tw = nullptr;
    cx<double>* d_twn = nullptr;
    cx<double>* d_twn_r = nullptr;   // [2][8][H/8], H = frame/2: W_{2NF}^(2j+P) in the register order of the DIF engine (if0_split_body)
    If0Wfir wf;
};

// frequency held by register e of thread t after dif_fft_keep_last<H>
template <int H>
static int if0_reg_freq_t(int t, int e) {
    constexpr int RL = DifPlan<H>::radix(DifPlan<H>::n - 1);
    return dif_freq<H>(dif_last_pos<H>(t, e / RL, e % RL));
}
static int if0_reg_freq(int H, int t, int e) {
    return H == 512 ? if0_reg_freq_t<512>(t, e) : (H == 1024 ? if0_reg_freq_t<1024>(t, e) : (H == 2048 ? if0_reg_freq_t<2048>(t, e) : if0_reg_freq_t<4096>(t, e)));
}

// Slowest pole radius of the per-channel chain (2 x resonator 1, 2 x resonator 2: radius A each; the 12 all-pass
// stages of the warped FIR: |a|; the Butterworth low-pass at the channel frequency: sqrt(a2)), computed from the SAME
// closed forms if0_plan uses, and the run-in it needs.  The four resonator sections of a channel have (to 1e-6) the same
// pole pair, so the response to what happened n samples ago is enveloped by n^3 rho^n, whose total is 6 / (1 - rho)^4; the
// part of it beyond W samples, relative to the whole, is the incomplete gamma ratio Q(4, u) = e^-u (u^3 + 3u^2 + 6u + 6) / 6
// at u = W (1 - rho).  W is the smallest multiple of 8192 (a multiple of every frame size), at least IF0_WARMUP, with
// Q(4, u) <= 1e-13: 40960 samples for the default chain (u = 43.8).  Measured on the restated reference (oracle,
// slowest channel, 22.05 and 44.1 kHz): starting 32768 samples early leaves 3e-12 .. 6e-12 of the sequential result,
// 40960 leaves 3e-15 .. 6e-15 -- the level at which the other channels differ between ANY two chunkings (5e-14: rounding
// through the rectifier).  (Rounds 1-2 bounded the ABSOLUTE envelope, rho^W W^3 <= 1e-15, which ignores the 1e-12 gain
// of the four sections and asked for 65536 samples: a third more front-end work per chunk.)
// 0 when the chain is too slow for IF0_WARMUP_MAX (or unstable).
// (c0, c1, min_w: the same for the channels [c0, c1) alone -- the wave of leftover channels has its own run-in, see
//  if0_run_host -- with a floor of min_w samples instead of IF0_WARMUP)
static long long if0_warmup_range(int fs, const mpx_if0_params& p, double* rho_out, int c0, int c1, long long min_w);
static long long if0_warmup(int fs, const mpx_if0_params& p, double* rho_out) {
    return if0_warmup_range(fs, p, rho_out, 0, p.channels, IF0_WARMUP);
}
static long long if0_warmup_range(int fs, const mpx_if0_params& p, double* rho_out, int c0, int c1, long long min_w) {
    double rho = std::fabs(1.0674 * std::sqrt((2.0 / M_PI) * std::atan(0.06583 * fs / 1000.0)) - 0.1916);
    for (int c = c0; c < c1; ++c) {
        const double fc = 229 * (std::pow(10.0, (p.zeta1 * c + p.zeta0) / 21.4) - 1);
        const double A = std::exp(-(3.0 / 4) * M_PI / (fc * std::sqrt(std::pow(2.0, 1.0 / 4) - 1)));   // quirk A.1: "fs" is fc
        const double kk = std::tan(M_PI * fc / fs);
        const double a2 = (1.0 - std::sqrt(2.0) * kk + kk * kk) / (1.0 + std::sqrt(2.0) * kk + kk * kk);
        rho = std::max(rho, std::max(A, std::sqrt(std::fabs(a2))));
    }
    if (rho_out) *rho_out = rho;
    if (!(rho < 1.0)) return 0;
    for (long long w = min_w; w <= IF0_WARMUP_MAX; w += 8192) {
        const double u = (double)w * (1.0 - rho);
        if (-u + std::log((u * u * u + 3.0 * u * u + 6.0 * u + 6.0) / 6.0) <= std::log(1e-13)) return w;
    }
    return 0;
}

int remez_taps_for(mpx_ctx* ctx, int fs, double* c13);  // mpx_esacf.hip

static void if0_host_fft(std::vector<cx<double>>& a) {  // in-place radix-2, forward; plan tables only
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
                const long double wx = cosl(ang), wy = sinl(ang);
                const cx<double> u = a[i + k], t = a[i + k + len / 2];
                const cx<double> v = {(double)(t.x * wx - t.y * wy), (double)(t.x * wy + t.y * wx)};
                a[i + k] = {u.x + v.x, u.y + v.y};
                a[i + k + len / 2] = {u.x - v.x, u.y - v.y};
            }
}

// Overlapped time slices (below, in if0_run_host): measured and not adopted -- development builds only, MPX_IF0_OVERLAP=1
static int if0_overlap_mode(mpx_ctx*) { return DEV_KNOBS ? dev_env_int("MPX_IF0_OVERLAP", 0) : 0; }
// the spectra's stream and the four hand-over events; false (and the call runs without overlap) if they cannot be had
static bool if0_overlap_ready(mpx_ctx* ctx) {
    if (ctx->if0_overlap_made) return ctx->if0_sp_stream != nullptr;
    ctx->if0_overlap_made = true;
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
    bool ok = hipStreamCreateWithPriority(&ctx->if0_sp_stream, hipStreamNonBlocking, least) == hipSuccess;
    for (int k = 0; ok && k < 2; ++k)
        ok = hipEventCreateWithFlags(&ctx->if0_ev_fe[k], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&ctx->if0_ev_sp[k], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        for (hipEvent_t* e : {&ctx->if0_ev_fe[0], &ctx->if0_ev_fe[1], &ctx->if0_ev_sp[0], &ctx->if0_ev_sp[1]}) {
            if (*e) (void)hipEventDestroy(*e);
            *e = nullptr;
        }
        if (ctx->if0_sp_stream) (void)hipStreamDestroy(ctx->if0_sp_stream);
        ctx->if0_sp_stream = nullptr;
    }
    return ok;
}

static int if0_plan(mpx_ctx* ctx, int fs, const mpx_if0_params& p, If0Plan& plan) {
    char keyb[256];
    snprintf(keyb, sizeof keyb, "if0r4_%d_%d_%d_%.17g_%.17g", fs, p.frame_size, p.channels, p.zeta0, p.zeta1);
    const std::string key = keyb;
    int rc = remez_taps_for(ctx, fs, plan.wf.c);
    if (rc) return rc;
    plan.wf.a = 1.0674 * std::sqrt((2.0 / M_PI) * std::atan(0.06583 * fs / 1000.0)) - 0.1916;
    auto it = ctx->misc_plans.find(key);
    if (it != ctx->misc_plans.end()) {
        plan.d_coefs = (If0ChanCoef*)it->second[0];
        plan.d_window = (double*)it->second[1];
        plan.d_tw = (cx<double>*)it->second[2];
        plan.d_twn = (cx<double>*)it->second[3];
        plan.d_twn_r = (cx<double>*)it->second[4];
        plan.d_one = (double*)it->second[5];
        return MPX_OK;
    }
    std::vector<If0ChanCoef> coefs(p.channels);
    for (int c = 0; c < p.channels; ++c) {
        const double fc = 229 * (std::pow(10.0, (p.zeta1 * c + p.zeta0) / 21.4) - 1);  // iterative_f0.py:37-39
        // _auditory_filterbank(x, fc, fs) is CALLED as (x, fs, fc) (quirk A.1): inside, "fc" is the sample
        // rate and "fs" is the channel frequency
        const double in_fc = (double)fs, in_fs = fc;
        const int J = 4;
        const double A = std::exp(-(3.0 / J) * M_PI / (in_fs * std::sqrt(std::pow(2.0, 1.0 / J) - 1)));
        const double cos_theta1 = (1 + A * A) / (2 * A) * std::cos(2 * M_PI * in_fc / in_fs);
        const double cos_theta2 = (2 * A) / (1 + A * A) * std::cos(2 * M_PI * in_fc / in_fs);
        const double rho1 = (1.0 / 2) * (1 - A * A);
        const double rho2 = (1 - A * A) * std::sqrt(1 - cos_theta2 * cos_theta2);
        If0ChanCoef k;
        k.r1b0 = rho1;
        k.r1b2 = -rho1;
        k.r1a1 = -A * cos_theta1;
        k.r1a2 = A * A;
        k.r2b0 = rho2;
        k.r2a1 = -A * cos_theta2;
        k.r2a2 = A * A;
        if (!(fc > 0.0 && fc < fs / 2.0))
            return set_error(ctx, MPX_EINVAL, "iterative F0: channel %d centre %.1f Hz is outside (0, fs/2): scipy's "
                             "butter raises ValueError here", c, fc);
        const double kk = std::tan(M_PI * fc / fs);  // lowpass_filter(yc, fs, fc)
        const double norm = 1.0 / (1.0 + std::sqrt(2.0) * kk + kk * kk);
        k.lpb0 = kk * kk * norm;
        k.lpb1 = 2.0 * kk * kk * norm;
        k.lpb2 = kk * kk * norm;
        k.lpa1 = 2.0 * (kk * kk - 1.0) * norm;
        k.lpa2 = (1.0 - std::sqrt(2.0) * kk + kk * kk) * norm;
        coefs[c] = k;
    }
    const int NF = p.frame_size;
    std::vector<double> win(NF);
    for (int i = 0; i < NF; ++i) win[i] = 0.54 - 0.46 * std::cos(2.0 * M_PI * i / (double)(NF - 1));
    // W_NF^j, j < NF, and W_2NF^k, k <= NF (mpx_if0_tables.hpp).  Chirp-z frame sizes build their own tables below and skip these.
    const bool tuned = NF == 1024 || NF == 2048 || NF == 4096 || NF == 8192;
    std::vector<cx<double>> tw(NF), twn(NF + 1);
    static_assert(sizeof(cx<double>) == 2 * sizeof(double), "cx<double> is (x, y)");
    if (tuned) {
        // (the tables depend on the frame size alone: a second context of the process, or a second sample rate, copies the first's)
        static std::mutex roots_mutex;
        static std::map<int, std::pair<std::vector<cx<double>>, std::vector<cx<double>>>> roots;
        std::lock_guard<std::mutex> lock(roots_mutex);
        auto it = roots.find(NF);
        if (it == roots.end()) {
            if0_unit_roots(NF, reinterpret_cast<double*>(tw.data()), reinterpret_cast<double*>(twn.data()));
            roots.emplace(NF, std::make_pair(tw, twn));
        } else {
            tw = it->second.first;
            twn = it->second.second;
        }
    }
    plan.d_coefs = (If0ChanCoef*)upload(ctx, coefs.data(), coefs.size() * sizeof(If0ChanCoef));
    plan.d_window = (double*)upload(ctx, win.data(), win.size() * sizeof(double));
    plan.d_tw = (cx<double>*)upload(ctx, tw.data(), tw.size() * sizeof(cx<double>));
    plan.d_twn = (cx<double>*)upload(ctx, twn.data(), twn.size() * sizeof(cx<double>));
    if (tuned) {
        const int H = NF / 2, T = H / 8;
        std::vector<cx<double>> tr((size_t)2 * H);
        for (int P = 0; P < 2; ++P)
            for (int e = 0; e < 8; ++e)
                for (int t = 0; t < T; ++t) tr[((size_t)P * 8 + e) * T + t] = twn[2 * if0_reg_freq(H, t, e) + P];
        plan.d_twn_r = (cx<double>*)upload(ctx, tr.data(), tr.size() * sizeof(cx<double>));
    } else {
        // Any other frame size (iterative_f0.py:25 takes any integer): the 2 NF-point spectrum by chirp-z on the padded
        // Stockham engine, X[k] = conj(c[k]) sum_n (x[n] conj(c[n])) c[k - n], c[m] = exp(i pi m^2 / 2NF), n < NF, k <= NF:
        // a cyclic convolution of L >= 2 NF points.  d_tw: W_L, d_twn: c[0 .. NF], d_twn_r: FFT_L(c on -(NF-1) .. NF) / L.
        // Above 4096 samples L = 16384 is two residues of 8192 points around one radix-2 step (if0_spectrum_blue2_kernel):
        // d_tw: W_8192, d_twn_r: the filter spectrum as [residue r][k] = bin 2 k + r, then W_16384^m for m < 8192.
        int L = 4096;
        while (L < 2 * NF) L <<= 1;
        const int LE = L > 8192 ? 8192 : L;   // points of the LDS engine
        tw.assign((size_t)LE, cx<double>{0.0, 0.0});
        for (int j = 0; j < LE; ++j) {
            const long double ang = -2.0L * M_PIl * j / (long double)LE;
            tw[(size_t)j] = {(double)cosl(ang), (double)sinl(ang)};
        }
        const long long n2 = 2LL * NF;
        for (long long m = 0; m <= NF; ++m) {
            const long long q = (m * m) % (2 * n2);   // exact phase reduction: pi m^2 / n2 = pi q / n2 (mod 2 pi)
            const long double ang = M_PIl * (long double)q / (long double)n2;
            twn[(size_t)m] = {(double)cosl(ang), (double)sinl(ang)};
        }
        std::vector<cx<double>> filt((size_t)L, cx<double>{0.0, 0.0});
        for (int m = 0; m <= NF; ++m) filt[(size_t)m] = twn[(size_t)m];
        for (int m = 1; m < NF; ++m) filt[(size_t)(L - m)] = twn[(size_t)m];
        if0_host_fft(filt);
        for (auto& v : filt) {
            v.x /= L;
            v.y /= L;
        }
        if (L > 16384) {   // four residues (if0_spectrum_blue4_kernel): [r][k] = bin 4 k + r, then W_L^{m r} for r = 1 .. 3, m < 8192
            std::vector<cx<double>> fr((size_t)7 * LE);
            for (int r = 0; r < 4; ++r)
                for (int k = 0; k < LE; ++k) fr[(size_t)r * LE + k] = filt[(size_t)4 * k + r];
            for (int r = 1; r < 4; ++r)
                for (int m = 0; m < LE; ++m) {
                    const long long e = ((long long)m * r) % L;   // exact phase reduction
                    const long double ang = -2.0L * M_PIl * (long double)e / (long double)L;
                    fr[(size_t)(3 + r) * LE + m] = {(double)cosl(ang), (double)sinl(ang)};
                }
            filt.swap(fr);
        } else if (L > 8192) {
            std::vector<cx<double>> fr((size_t)3 * LE);
            for (int r = 0; r < 2; ++r)
                for (int k = 0; k < LE; ++k) fr[(size_t)r * LE + k] = filt[(size_t)2 * k + r];
            for (int m = 0; m < LE; ++m) {
                const long double ang = -2.0L * M_PIl * m / (long double)L;
                fr[(size_t)2 * LE + m] = {(double)cosl(ang), (double)sinl(ang)};
            }
            filt.swap(fr);
        }
        hipFree(plan.d_tw);    // (the NF-point tables uploaded above are of no use to this path)
        hipFree(plan.d_twn);
        for (void* dead : {(void*)plan.d_tw, (void*)plan.d_twn})
            ctx->owned.erase(std::remove(ctx->owned.begin(), ctx->owned.end(), dead), ctx->owned.end());
        plan.d_tw = (cx<double>*)upload(ctx, tw.data(), tw.size() * sizeof(cx<double>));
        plan.d_twn = (cx<double>*)upload(ctx, twn.data(), twn.size() * sizeof(cx<double>));
        plan.d_twn_r = (cx<double>*)upload(ctx, filt.data(), filt.size() * sizeof(cx<double>));
    }
    const double one = 1.0;
    plan.d_one = (double*)upload(ctx, &one, sizeof one);
    if (!plan.d_coefs || !plan.d_window || !plan.d_tw || !plan.d_twn || !plan.d_twn_r || !plan.d_one) return MPX_ENOMEM;
    ctx->misc_plans[key] = {plan.d_coefs, plan.d_window, plan.d_tw, plan.d_twn, plan.d_twn_r, plan.d_one};
    return MPX_OK;
}

// The same spectrum on the in-place DIF engine (mpx_fft_dif.hpp).  The 2*NF-point frame is zero in its upper
// half, so the NF-point complex sequence z[m] = x[2m] + i x[2m+1] is zero for m >= NF/2 and its DFT splits
// without arithmetic: Z[2j] = DFT_H(z)[j], Z[2j+1] = DFT_H(z W_NF^m)[j], H = NF/2 -- two H-point transforms
// (H <= 4096: wave-local passes, one barrier each) instead of one padded NF-point Stockham transform, 64 KB of LDS
// instead of 139 KB (two workgroups per CU).  Outputs stay digit-reversed in registers: |X|^power is summed over
// the channels per bin, and a sum does not care which register holds which bin.
template <int NF, int P>   // P = 0: even bins 2j (and bin NF), P = 1: odd bins 2j + 1
__device__ __forceinline__ void if0_half_spectrum(cx<double>* buf, const DifTwiddles<NF / 2, double>& twd, const double* __restrict__ src,
                                                  int valid, const double* __restrict__ window,
                                                  const cx<double>* __restrict__ twNF, const cx<double>* __restrict__ twn,
                                                  double power, double* acc, double& acc_nyq) {
    constexpr int H = NF / 2, T = H / 8;
    using PL = DifPlan<H>;
    constexpr int RL = PL::radix(PL::n - 1);
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));  // nothing below may be hoisted out of the caller's channel loop (registers)
    cx<double> regs[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int m = tid + r * T, s0 = 2 * m;
        double x0 = 0.0, x1 = 0.0;
        if (s0 + 1 < valid) {
            const cx<double> v = *reinterpret_cast<const cx<double>*>(src + s0);   // (windowed by the front end)
            x0 = v.x;
            x1 = v.y;
        } else if (s0 < valid) {
            x0 = src[s0];
        }
        regs[r] = {x0, x1};
        if (P) regs[r] = cmul(regs[r], twNF[m]);
    }
    dif_fft_keep_last<H, double>(buf, twd, regs, tid);
#pragma unroll
    for (int e = 0; e < 8; ++e) buf[sigma<H>(dif_last_pos<H>(tid, e / RL, e % RL))] = regs[e];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int j = dif_freq<H>(dif_last_pos<H>(tid, e / RL, e % RL));
        const int jm = P ? H - 1 - j : (H - j) & (H - 1);   // Z[NF - k] lives in the same half-transform
        const cx<double> A = regs[e];
        cx<double> B = buf[sigma<H>(dif_pos<H>(jm))];
        B.y = -B.y;
        const cx<double> E = {0.5 * (A.x + B.x), 0.5 * (A.y + B.y)};
        const cx<double> D = {0.5 * (A.x - B.x), 0.5 * (A.y - B.y)};
        const cx<double> X = cadd(E, mul_mi(cmul(twn[2 * j + P], D)));
        const double mag = sqrt(X.x * X.x + X.y * X.y);  // |X| of audio-range data: no need for hypot's scaling
        acc[e] += power == 1.0 ? mag : pow(mag, power);
        if (!P && j == 0) {  // bin NF pairs Z[0] with itself
            const cx<double> Xn = cadd(E, mul_mi(cmul(twn[NF], D)));
            const double mn = sqrt(Xn.x * Xn.x + Xn.y * Xn.y);
            acc_nyq += power == 1.0 ? mn : pow(mn, power);
        }
    }
    __syncthreads();  // the mirror reads are done before the next transform writes buf
}

#ifdef MPX_DEV_KNOBS
template <int NF>
__global__ __launch_bounds__(NF / 16, 4) void if0_spectrum_dif_kernel(const double* __restrict__ yc, const If0Frame* __restrict__ frames,
                                                                     int channels, double power, const double* __restrict__ window,
                                                                     const cx<double>* __restrict__ twNF,  // W_NF^j, j < NF
                                                                     const cx<double>* __restrict__ twn,   // W_{2NF}^k, k <= NF
                                                                     double* __restrict__ ut) {            // [F, 2*NF]
    constexpr int H = NF / 2;
    using PL = DifPlan<H>;
    constexpr int RL = PL::radix(PL::n - 1);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cx<double>* buf = reinterpret_cast<cx<double>*>(smem);
    const int tid = threadIdx.x;
    const If0Frame fr = frames[blockIdx.x];
    DifTwiddles<H, double> twd;
#pragma unroll
    for (int i = 0; i < PL::n - 1; ++i) twd.w[i] = twNF[2 * ((tid & (PL::stride(i) - 1)) * (H / PL::block(i)))];  // W_H = W_NF^2
    double acc_e[8], acc_o[8], acc_nyq = 0.0;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc_e[e] = acc_o[e] = 0.0;
    for (int ch = 0; ch < channels; ++ch) {
        const double* src = yc + fr.yc_base + (size_t)ch * fr.ch_stride;
        if0_half_spectrum<NF, 0>(buf, twd, src, fr.valid, window, twNF, twn, power, acc_e, acc_nyq);
        if0_half_spectrum<NF, 1>(buf, twd, src, fr.valid, window, twNF, twn, power, acc_o, acc_nyq);
    }
    double* row = ut + (size_t)blockIdx.x * 2 * NF;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int j = dif_freq<H>(dif_last_pos<H>(tid, e / RL, e % RL));
        const int k0 = 2 * j, k1 = 2 * j + 1;
        row[k0] = acc_e[e];
        if (k0 > 0) row[2 * NF - k0] = acc_e[e];  // |X[N-k]| = |X[k]| for a real frame
        row[k1] = acc_o[e];
        row[2 * NF - k1] = acc_o[e];
        if (j == 0) row[NF] = acc_nyq;
    }
}

#endif  // MPX_DEV_KNOBS

// Round 3: the same transform pair, ONE parity per workgroup.  A workgroup that computes both parities carries 17
// accumulators across the channel loop next to two transforms' worth of temporaries: under the 128 registers that two
// workgroups per CU allow, 184 bytes per lane went to scratch, and each parity fetched the frame's samples again -- measured
// HBM traffic 3.2 x the front end's output.  Split by parity, a workgroup keeps 8 accumulators (+ the Nyquist bin),
// prefetches the next channel's samples under the current transform and stays in registers; the two workgroups of a frame
// sit on the same XCD (block ids 8 apart) and start together, so the second read of a frame's samples is an L2 / MALL hit.
// sqrt of a sum of squares of audio-range data: zero, or far inside the normal range (>= 2^-767).  The compiler's own
// expansion of sqrt(double) -- v_rsq_f64 and these Newton steps -- wrapped in a scaling of tiny arguments (a compare, two
// selects, two v_ldexp) and a class test for 0 / inf: six of its eighteen instructions, eight times per channel and thread.
// Same steps on the same operands: the same bits wherever the scaling would not have engaged.
__device__ __forceinline__ double sqrt_sumsq(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return x == 0.0 ? 0.0 : g;
}

// d * W_16^E, E a compile-time exponent < 8 (W = exp(-2 pi i / 16))
template <int E>
__device__ __forceinline__ cx<double> if0_mul_w16(cx<double> d) {
    constexpr double H2 = 0.70710678118654752440, C1 = 0.92387953251128675613, S1 = 0.38268343236508977173;
    if constexpr (E == 0) return d;
    else if constexpr (E == 4) return {d.y, -d.x};
    else if constexpr (E == 2) return {H2 * (d.x + d.y), H2 * (d.y - d.x)};
    else if constexpr (E == 6) return {H2 * (d.y - d.x), -H2 * (d.x + d.y)};
    else {
        constexpr double c = E == 1 ? C1 : (E == 3 ? S1 : (E == 5 ? -S1 : -C1)), sn = (E == 1 || E == 7) ? S1 : C1;   // W = c - i sn
        return {d.x * c + d.y * sn, d.y * c - d.x * sn};
    }
}
__device__ __forceinline__ cx<double> if0_mul_w16_e(cx<double> d, int e) {   // e is a constant after unrolling
    switch (e) {
        case 0: return if0_mul_w16<0>(d);
        case 1: return if0_mul_w16<1>(d);
        case 2: return if0_mul_w16<2>(d);
        case 3: return if0_mul_w16<3>(d);
        case 4: return if0_mul_w16<4>(d);
        case 5: return if0_mul_w16<5>(d);
        case 6: return if0_mul_w16<6>(d);
        default: return if0_mul_w16<7>(d);
    }
}
template <int NF, int P, bool POW1, int IF0_PF, bool FULL>   // FULL: every sample of the frame exists (no zero padding to select)
__device__ __forceinline__ void if0_split_body(cx<double>* buf, const double* __restrict__ yc, const If0Frame fr, int channels,
                                               double power, const double* __restrict__ window, const cx<double>* __restrict__ twNF,
                                               const cx<double>* __restrict__ twn, const cx<double>* __restrict__ twn_r,
                                               double* __restrict__ row) {
    constexpr int H = NF / 2, T = H / 8;
    using PL = DifPlan<H>;
    constexpr int RL = PL::radix(PL::n - 1);
    const int tid0 = threadIdx.x;
    DifTwiddles<H, double> twd;
#pragma unroll
    for (int i = 0; i < PL::n - 1; ++i) twd.w[i] = twNF[2 * ((tid0 & (PL::stride(i) - 1)) * (H / PL::block(i)))];  // W_H = W_NF^2
    // NF = 8192 (16 T): the modulation W_NF^m of the odd parity, m = tid + r T, is W_NF^tid x W_16^r, and the split twiddle of
    // register e, W_2NF^(2 (j0 + e H/8) + P), is W_2NF^(2 j0 + P) x W_16^e -- one table value per thread and eight constants
    // where every channel read sixteen table values per thread (as much as its samples).  (The two values are re-read per
    // channel: held across the loop they cost the eight registers this kernel does not have under four waves per SIMD.)
    constexpr bool TW16 = NF == 8192 || NF == 1024;   // (the sizes whose last DIF pass is a radix-8 one)
    constexpr bool PAIR = TW16 && POW1;   // one thread splits a bin AND its mirror bin (below); with pow() inlined twice per pair the kernel spilled
    double acc[8];
    // the Nyquist bin belongs to the one thread that holds bin 0: its sum lives in LDS (touched by that thread alone) instead
    // of in two registers of every thread
    __shared__ double nyq_sh;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        acc[e] = 0.0;
        if (!P && dif_freq<H>(dif_last_pos<H>(tid0, e / RL, e % RL)) == 0) nyq_sh = 0.0;
    }
    // samples 2m, 2m + 1 of the frame, m = tid + r T.  The front end writes whole frames (the filters ring on past the end
    // of a clip), the reference pads the FILTERED signal with zeros: loads are unconditional, samples from `valid` on are
    // replaced by zeros (selects, no branches).
    const double* src = yc + fr.yc_base;
    cx<double> xn[8];
    // (the samples arrive Hamming-windowed: the front end's tile flush multiplies them, see if0_frontend_body)
    // FULL: plain 16-byte loads.  Otherwise (round 6) through a buffer descriptor of exactly the samples that exist -- base = the
    // channel's row, size = 8 x valid bytes: a sample from `valid` on comes back as zero from the hardware's range check, one
    // 8-byte load per sample so that a load is either inside or outside (no compare, no select: those were 8 % of the kernel's
    // vector instructions and kept the body of a clip's last frame from fetching ahead).
    auto fetch = [&](const double* __restrict__ p, int tid) {
        if constexpr (FULL) {
#pragma unroll
            for (int r = 0; r < 8; ++r) xn[r] = *reinterpret_cast<const cx<double>*>(p + 2 * (tid + r * T));
        } else {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p), 0, fr.valid > 0 ? 8 * fr.valid : 0, 0x00020000);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int voff = 16 * (tid + r * T);
                const auto lo = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, 0);
                const auto hi = __builtin_amdgcn_raw_buffer_load_b64(rs, voff + 8, 0, 0);
                xn[r] = {__builtin_bit_cast(double, lo), __builtin_bit_cast(double, hi)};
            }
        }
    };
    fetch(src, tid0);
    for (int ch = 0; ch < channels; ++ch) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));  // nothing below may be hoisted out of the channel loop (registers)
        cx<double> regs[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int m = tid + r * T;
            regs[r] = xn[r];
            if (P && !TW16) regs[r] = cmul(regs[r], twNF[m]);
        }
        if (P && TW16) {
            const cx<double> wmod = twNF[tid];
#pragma unroll
            for (int r = 0; r < 8; ++r) regs[r] = if0_mul_w16_e(cmul(regs[r], wmod), r);
        }
        if (IF0_PF == 2)
        // the next channel's samples travel under this channel's transform (the last channel re-reads itself)
        fetch(src + (size_t)(ch + 1 < channels ? ch + 1 : ch) * fr.ch_stride, tid);
        // (the barrier behind the previous channel's mirror reads sits inside, between the first pass's arithmetic and its stores)
        dif_fft_keep_last_barrier_before_stores<H, double>(buf, twd, regs, tid);
#pragma unroll
        for (int e = PAIR ? 4 : 0; e < 8; ++e) buf[sigma<H>(dif_last_pos<H>(tid, e / RL, e % RL))] = regs[e];
        if (IF0_PF == 1)
        fetch(src + (size_t)(ch + 1 < channels ? ch + 1 : ch) * fr.ch_stride, tid);
        const cx<double> wsplit = TW16 ? twn[2 * dif_freq<H>(dif_last_pos<H>(tid, 0, 0)) + P] : cx<double>{1.0, 0.0};
        __syncthreads();
        if constexpr (PAIR) {
            // A bin and its mirror bin come from the same pair (Z[j], Z[jm]): with t = W^k D they are E - i t and E + i t.  A thread's
            // bins are j0 + e H/8, their mirrors (H - 1 - j0) + (7 - e) H/8 (odd parity) or (H/8 - j0) + (7 - e) H/8 (even parity,
            // j0 > 0): register 7 - e of ONE partner thread.  So a thread splits its registers 0..3 against the partner's 7..4 and
            // takes BOTH bins of each pair -- four LDS writes and four reads per thread where every thread wrote and read eight,
            // one twiddle product per pair.  Even parity, j0 = 0 (one thread): Z[0] pairs with itself, its "mirror" bin is NF
            // (Nyquist), and Z[H/2] -- register 4 -- is its own mirror: that bin is the thread's extra one (nyq_sh).
            const int j0 = dif_freq<H>(dif_last_pos<H>(tid, 0, 0));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = j0 + e * (H / 8);
                const int jm = P ? H - 1 - j : (H - j) & (H - 1);
                const cx<double> A = regs[e];
                cx<double> B = buf[sigma<H>(dif_pos<H>(jm))];
                if (!P && j == 0) B = A;   // (what was read there is stale: register 0 is not exchanged)
                B.y = -B.y;
                const cx<double> E = {0.5 * (A.x + B.x), 0.5 * (A.y + B.y)};
                const cx<double> D = {0.5 * (A.x - B.x), 0.5 * (A.y - B.y)};
                const cx<double> t = if0_mul_w16_e(cmul(wsplit, D), e);
                const double x0 = E.x + t.y, y0 = E.y - t.x, x1 = E.x - t.y, y1 = E.y + t.x;
                const double m0 = sqrt_sumsq(x0 * x0 + y0 * y0), m1 = sqrt_sumsq(x1 * x1 + y1 * y1);
                acc[e] += POW1 ? m0 : pow(m0, power);
                acc[4 + e] += POW1 ? m1 : pow(m1, power);
            }
            if (!P && j0 == 0) {
                const cx<double> A = regs[4];   // Z[H/2] and its own conjugate: E = (A.x, 0), D = (0, A.y)
                const cx<double> t = if0_mul_w16_e(cmul(wsplit, cx<double>{0.0, A.y}), 4);
                const double x0 = A.x + t.y, y0 = -t.x;
                const double m0 = sqrt_sumsq(x0 * x0 + y0 * y0);
                nyq_sh += POW1 ? m0 : pow(m0, power);
            }
        } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int j = dif_freq<H>(dif_last_pos<H>(tid, e / RL, e % RL));
            const int jm = P ? H - 1 - j : (H - j) & (H - 1);   // Z[NF - k] lives in the same half-transform
            const cx<double> A = regs[e];
            cx<double> B = buf[sigma<H>(dif_pos<H>(jm))];
            B.y = -B.y;
            const cx<double> E = {0.5 * (A.x + B.x), 0.5 * (A.y + B.y)};
            const cx<double> D = {0.5 * (A.x - B.x), 0.5 * (A.y - B.y)};
            // split twiddle W_{2NF}^(2j+P) from the table in register order (by bin it was a 16-byte gather per lane)
            const cx<double> X = cadd(E, mul_mi(TW16 ? if0_mul_w16_e(cmul(wsplit, D), e) : cmul(twn_r[(P * 8 + e) * T + tid], D)));
            const double mag = sqrt_sumsq(X.x * X.x + X.y * X.y);  // |X| of audio-range data: no need for hypot's scaling
            acc[e] += POW1 ? mag : pow(mag, power);
            if (!P && j == 0) {  // bin NF pairs Z[0] with itself
                const cx<double> Xn = cadd(E, mul_mi(cmul(twn[NF], D)));
                const double mn = sqrt_sumsq(Xn.x * Xn.x + Xn.y * Xn.y);
                nyq_sh += POW1 ? mn : pow(mn, power);
            }
        }
        }
        if (IF0_PF == 0)
        fetch(src + (size_t)(ch + 1 < channels ? ch + 1 : ch) * fr.ch_stride, tid);
    }
    if constexpr (PAIR) {
        const int j0 = dif_freq<H>(dif_last_pos<H>(tid0, 0, 0));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = 2 * (j0 + e * (H / 8)) + P, km = NF - k;   // the bin and its mirror bin
            row[k] = acc[e];
            if (k > 0) row[2 * NF - k] = acc[e];  // |X[N-k]| = |X[k]| for a real frame
            row[km] = acc[4 + e];
            if (km < NF) row[2 * NF - km] = acc[4 + e];
        }
        if (!P && j0 == 0) {   // the bin that is its own mirror
            row[NF / 2] = nyq_sh;
            row[2 * NF - NF / 2] = nyq_sh;
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int j = dif_freq<H>(dif_last_pos<H>(tid0, e / RL, e % RL));
        const int k = 2 * j + P;
        row[k] = acc[e];
        if (k > 0) row[2 * NF - k] = acc[e];  // |X[N-k]| = |X[k]| for a real frame
        if (!P && j == 0) row[NF] = nyq_sh;
    }
}

// Waves per SIMD the register allocator is held to: four at the default frame size with power 1 (121-124 registers, no
// scratch) and at 1024; the other shapes a caller can reach -- frames of 2048 / 4096, |X|^power through pow() -- spilled
// 96-324 bytes per lane under that limit (round 3) and are scratch-free at two or three waves (tests/test_kernel_resources.py).
__host__ __device__ constexpr int if0_split_waves(int NF, bool POW1) {
    return POW1 ? ((NF == 8192 || NF == 1024) ? 4 : 2) : (NF == 8192 ? 3 : 2);
}
template <int NF, bool POW1, int PF>
__global__ __launch_bounds__(NF / 16, if0_split_waves(NF, POW1)) void if0_spectrum_split_kernel(const double* __restrict__ yc, const If0Frame* __restrict__ frames,
                                                                       long long nframes, int channels, double power,
                                                                       const double* __restrict__ window,
                                                                       const cx<double>* __restrict__ twNF,  // W_NF^j, j < NF
                                                                       const cx<double>* __restrict__ twn,   // W_{2NF}^k, k <= NF
                                                                       const cx<double>* __restrict__ twn_r, // the same, register order
                                                                       double* __restrict__ ut) {            // [F, 2*NF]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cx<double>* buf = reinterpret_cast<cx<double>*>(smem);
    // block b = 16 q + 8 P + r  ->  frame 8 q + r, parity P: the two workgroups of a frame are 8 block ids apart (same XCD)
    const long long b = blockIdx.x;
    const long long f = (b >> 4) * 8 + (b & 7);
    const int P = (int)(b >> 3) & 1;
    if (f >= nframes) return;
    const If0Frame fr = frames[f];
    double* row = ut + (size_t)f * 2 * NF;
    // (most frames are whole -- all but the last of a stream, five of the six of a two-second clip: their 32 selects and 16
    //  compares per channel are 8 % of this kernel's vector instructions)
    // (only for the default frame size, whose instantiation stays scratch-free with four bodies: the others keep one per parity)
    if (NF != 8192) {
        if (P) if0_split_body<NF, 1, POW1, PF, false>(buf, yc, fr, channels, power, window, twNF, twn, twn_r, row);
        else if0_split_body<NF, 0, POW1, PF, false>(buf, yc, fr, channels, power, window, twNF, twn, twn_r, row);
    } else if (fr.valid >= NF) {
        if (P) if0_split_body<NF, 1, POW1, PF, true>(buf, yc, fr, channels, power, window, twNF, twn, twn_r, row);
        else if0_split_body<NF, 0, POW1, PF, true>(buf, yc, fr, channels, power, window, twNF, twn, twn_r, row);
    } else {
        // (the last frame of a clip: the same schedule, its samples through a buffer descriptor -- if0_split_body's fetch)
        if (P) if0_split_body<NF, 1, POW1, PF, false>(buf, yc, fr, channels, power, window, twNF, twn, twn_r, row);
        else if0_split_body<NF, 0, POW1, PF, false>(buf, yc, fr, channels, power, window, twNF, twn, twn_r, row);
    }
}

// Frame sizes that are not 1024 / 2048 / 4096 / 8192: one workgroup per frame, per channel a chirp-z transform of the
// Hamming-windowed frame on the padded Stockham engine (two L-point transforms: L = 4096 up to 2048 samples, 8192 up to
// 4095), |X[k]| = |y[k]| (the final chirp has modulus one), k <= NF, mirrored into the 2 NF-bin row like the other kernels.
// Correct and complete, not tuned: the reference's default and the powers of two around it run on the split kernel above.
template <int L, int T>
__global__ __launch_bounds__(T) void if0_spectrum_blue_kernel(const double* __restrict__ yc, const If0Frame* __restrict__ frames,
                                                              int NF, int channels, double power,
                                                              const double* __restrict__ window, const cx<double>* __restrict__ tw,
                                                              const cx<double>* __restrict__ chirp, const cx<double>* __restrict__ bhat,
                                                              double* __restrict__ ut) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cx<double>* buf = reinterpret_cast<cx<double>*>(smem);
    const int tid = threadIdx.x;
    const If0Frame fr = frames[blockIdx.x];
    constexpr int KPT = L / 2 / T + 1;   // bins k = tid + j T <= NF <= L / 2
    double acc[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) acc[j] = 0.0;
    cx<double> regs[L / T];
    const double* src = yc + fr.yc_base;
    for (int ch = 0; ch < channels; ++ch) {
        const double* x = src + (size_t)ch * fr.ch_stride;
        for (int n = tid; n < L; n += T) {
            cx<double> v = {0.0, 0.0};
            if (n < NF) {
                const double xv = n < fr.valid ? x[n] : 0.0;   // the reference pads the FILTERED signal with zeros
                const double xw = xv * window[n];
                const cx<double> c = chirp[n];
                v = {xw * c.x, -(xw * c.y)};
            }
            buf[lds_slot(n)] = v;
        }
        __syncthreads();
        fft_lds<L, T, false, double>(buf, tw, regs, tid);
        for (int k = tid; k < L; k += T) {   // swapped: the forward transform of (im, re) is the swapped inverse transform
            const cx<double> v = cmul(buf[lds_slot(k)], bhat[k]);
            buf[lds_slot(k)] = {v.y, v.x};
        }
        __syncthreads();
        fft_lds<L, T, false, double>(buf, tw, regs, tid);
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const int k = tid + j * T;
            if (k <= NF) {
                const cx<double> y = buf[lds_slot(k)];
                const double mag = hypot(y.x, y.y);
                acc[j] += power == 1.0 ? mag : pow(mag, power);
            }
        }
        __syncthreads();   // buf is rewritten by the next channel
    }
    double* row = ut + (size_t)blockIdx.x * 2 * NF;
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const int k = tid + j * T;
        if (k <= NF) {
            row[k] = acc[j];
            if (k > 0 && k < NF) row[2 * NF - k] = acc[j];   // |X[N-k]| = |X[k]| for a real frame
        }
    }
}

// 4097 ... 8191 samples: the convolution has 16384 points -- two residues of 8192 around one radix-2 step that stays in the
// workgroup (as sacf_huge_kernel, mpx_esacf.hip): forward by decimation in frequency (the input is zero from 8192 on, so
// u_r[m] = a[m] W_16384^{m r}), the filter spectrum per residue, inverse by decimation in time: y[k] = v_0[k] + conj(W_16384^k)
// v_1[k] for k <= NF < 8192; v_0 waits in a per-workgroup row in HBM while residue 1 is transformed (in registers it spilled:
// 456 bytes per lane; the grid is one persistent workgroup per CU, the rows stay in L2).  Four 8192-point transforms per channel.
template <int T>
__global__ __launch_bounds__(T) void if0_spectrum_blue2_kernel(const double* __restrict__ yc, const If0Frame* __restrict__ frames,
                                                               int NF, int channels, double power,
                                                               const double* __restrict__ window, const cx<double>* __restrict__ tw,
                                                               const cx<double>* __restrict__ chirp, const cx<double>* __restrict__ bhat_r,
                                                               double* __restrict__ ut, long long nframes, cx<double>* v0_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int M = 8192;   // bins k <= NF < M
    cx<double>* buf = reinterpret_cast<cx<double>*>(smem);
    const cx<double>* __restrict__ twL = bhat_r + 2 * M;
    const int tid = threadIdx.x;
    cx<double>* v0 = v0_rows + (size_t)blockIdx.x * M;   // element k: written and read by the same thread
  for (long long f = blockIdx.x; f < nframes; f += gridDim.x) {
    const If0Frame fr = frames[f];
    double* row = ut + (size_t)f * 2 * NF;   // the sums over the channels accumulate in the row itself (a bin belongs to one thread)
    for (int k = tid; k <= NF; k += T) row[k] = 0.0;
    cx<double> regs[M / T];
    const double* src = yc + fr.yc_base;
    for (int ch = 0; ch < channels; ++ch) {
        const double* x = src + (size_t)ch * fr.ch_stride;
#pragma nounroll   // (unrolled, four inlined transforms' hoisted twiddle loads spill 500 bytes per lane)
        for (int r = 0; r < 2; ++r) {
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));   // nothing below is hoisted out of the loops (twiddle and table addresses: registers)
            for (int n = tid; n < M; n += T) {
                cx<double> v = {0.0, 0.0};
                if (n < NF) {
                    const double xv = n < fr.valid ? x[n] : 0.0;
                    const double xw = xv * window[n];
                    const cx<double> c = chirp[n];
                    v = {xw * c.x, -(xw * c.y)};
                    if (r) v = cmul(v, twL[n]);
                }
                buf[lds_slot(n)] = v;
            }
            __syncthreads();
            fft_lds<M, T, false, double>(buf, tw, regs, tid);
            for (int k = tid; k < M; k += T) {   // swapped: the forward transform of (im, re) is the swapped inverse transform
                const cx<double> v = cmul(buf[lds_slot(k)], bhat_r[r * M + k]);
                buf[lds_slot(k)] = {v.y, v.x};
            }
            __syncthreads();
            fft_lds<M, T, false, double>(buf, tw, regs, tid);
            for (int k = tid; k <= NF; k += T) {
                const cx<double> s = buf[lds_slot(k)];
                const cx<double> v = {s.y, s.x};
                if (r == 0) {
                    v0[k] = v;
                } else {
                    const cx<double> w = twL[k], u = v0[k];   // v conj(w)
                    const double yr = u.x + (v.x * w.x + v.y * w.y), yi = u.y + (v.y * w.x - v.x * w.y);
                    const double mag = hypot(yr, yi);
                    row[k] += power == 1.0 ? mag : pow(mag, power);
                }
            }
            __syncthreads();   // buf is rewritten by the next residue / channel
        }
    }
    for (int k = tid + 1; k < NF; k += T) row[2 * NF - k] = row[k];   // |X[N-k]| = |X[k]| for a real frame
  }
}

// 8193 ... 16384 samples (round 6): the convolution has 32768 points -- FOUR residues of 8192 around a radix-4 step that stays in
// the workgroup.  Forward by decimation in frequency: the input a[n] (windowed frame times the conjugate chirp) is zero from
// NF <= 2 M on, so residue r of the spectrum, A[4 j + r], is the M-point transform of u_r[m] = (a[m] + (-i)^r a[m + M]) W_L^{m r};
// the filter spectrum per residue; inverse by decimation in time: y[k + M q] = sum_r i^{q r} conj(W_L^{k r}) v_r[k] for the
// k + M q <= NF (q <= 2: the one bin 2 M = NF at NF = 16384).  The partial sums wait in per-workgroup rows in HBM (element k of
// row q: written and read by the same thread; the grid is one persistent workgroup per CU, the rows stay in L2).  Eight
// 8192-point transforms per channel: correct, untuned, like every chirp-z frame size.
template <int T>
__global__ __launch_bounds__(T) void if0_spectrum_blue4_kernel(const double* __restrict__ yc, const If0Frame* __restrict__ frames,
                                                               int NF, int channels, double power,
                                                               const double* __restrict__ window, const cx<double>* __restrict__ tw,
                                                               const cx<double>* __restrict__ chirp, const cx<double>* __restrict__ bhat_r,
                                                               double* __restrict__ ut, long long nframes, cx<double>* acc_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int M = 8192, R = 4;
    cx<double>* buf = reinterpret_cast<cx<double>*>(smem);
    const cx<double>* __restrict__ twL = bhat_r + (size_t)R * M;   // [r - 1][m] = W_L^{m r}, r = 1 .. 3, L = 4 M
    cx<double>* accq = acc_rows + (size_t)blockIdx.x * 3 * M;      // [q][k], q = 0 .. 2
  for (long long f = blockIdx.x; f < nframes; f += gridDim.x) {
    const If0Frame fr = frames[f];
    double* row = ut + (size_t)f * 2 * NF;   // the sums over the channels accumulate in the row itself (a bin belongs to one thread)
    for (int k = threadIdx.x; k <= NF; k += T) row[k] = 0.0;
    cx<double> regs[M / T];
    const double* src = yc + fr.yc_base;
    for (int ch = 0; ch < channels; ++ch) {
        const double* x = src + (size_t)ch * fr.ch_stride;
#pragma nounroll
        for (int r = 0; r < R; ++r) {
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));   // nothing below is hoisted out of the loops (twiddle and table addresses: registers)
            for (int m = tid; m < M; m += T) {
                // a[m] and a[m + M]: (x w)[n] conj(chirp[n]), zero from NF on
                cx<double> a0 = {0.0, 0.0}, a1 = {0.0, 0.0};
                if (m < NF) {
                    const double xw = (m < fr.valid ? x[m] : 0.0) * window[m];
                    const cx<double> c = chirp[m];
                    a0 = {xw * c.x, -(xw * c.y)};
                }
                if (m + M < NF) {
                    const double xw = (m + M < fr.valid ? x[m + M] : 0.0) * window[m + M];
                    const cx<double> c = chirp[m + M];
                    a1 = {xw * c.x, -(xw * c.y)};
                }
                // (-i)^r a1: r = 1: (y, -x), 2: (-x, -y), 3: (-y, x)
                const cx<double> b1 = r == 0 ? a1 : (r == 1 ? cx<double>{a1.y, -a1.x} : (r == 2 ? cx<double>{-a1.x, -a1.y} : cx<double>{-a1.y, a1.x}));
                cx<double> v = {a0.x + b1.x, a0.y + b1.y};
                if (r) v = cmul(v, twL[(size_t)(r - 1) * M + m]);
                buf[lds_slot(m)] = v;
            }
            __syncthreads();
            fft_lds<M, T, false, double>(buf, tw, regs, tid);
            for (int k = tid; k < M; k += T) {   // swapped: the forward transform of (im, re) is the swapped inverse transform
                const cx<double> v = cmul(buf[lds_slot(k)], bhat_r[(size_t)r * M + k]);
                buf[lds_slot(k)] = {v.y, v.x};
            }
            __syncthreads();
            fft_lds<M, T, false, double>(buf, tw, regs, tid);
            for (int k = tid; k < M; k += T) {
                const cx<double> sw = buf[lds_slot(k)];
                cx<double> v = {sw.y, sw.x};
                if (r) {   // conj(W_L^{k r}) v
                    const cx<double> w = twL[(size_t)(r - 1) * M + k];
                    v = {v.x * w.x + v.y * w.y, v.y * w.x - v.x * w.y};
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    if ((long long)k + (long long)M * q <= NF) {
                        // i^{q r} v
                        const int e = (q * r) & 3;
                        const cx<double> t = e == 0 ? v : (e == 1 ? cx<double>{-v.y, v.x} : (e == 2 ? cx<double>{-v.x, -v.y} : cx<double>{v.y, -v.x}));
                        cx<double> acc = r == 0 ? cx<double>{0.0, 0.0} : accq[(size_t)q * M + k];
                        acc = {acc.x + t.x, acc.y + t.y};
                        if (r + 1 < R) {
                            accq[(size_t)q * M + k] = acc;
                        } else {
                            const double mag = hypot(acc.x, acc.y);
                            row[k + M * q] += power == 1.0 ? mag : pow(mag, power);
                        }
                    }
                }
            }
            __syncthreads();   // buf is rewritten by the next residue / channel
        }
    }
    for (int k = threadIdx.x + 1; k < NF; k += T) row[2 * NF - k] = row[k];   // |X[N-k]| = |X[k]| for a real frame
  }
}

static int if0_spectrum_blue_launch(mpx_ctx* ctx, const double* yc, const If0Frame* frames, long long nf, int NF, int channels,
                                    double power, const If0Plan& plan, double* ut, hipStream_t st) {
    if (2 * NF <= 4096) {
        constexpr int L = 4096, T = 256;
        const size_t lds = sizeof(cx<double>) * lds_slots(L);
        auto kern = if0_spectrum_blue_kernel<L, T>;
        MPX_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)nf), dim3(T), lds, st, yc, frames, NF, channels, power, plan.d_window, plan.d_tw,
                           plan.d_twn, plan.d_twn_r, ut);
    } else if (2 * NF > 16384) {   // 8193 ... 16384 samples: four residues
        constexpr int T = 512;
        const size_t lds = sizeof(cx<double>) * lds_slots(8192);
        const long long grid = nf < ctx->num_cus ? nf : ctx->num_cus;   // persistent: one workgroup per CU (139 KB of LDS)
        int rc = ensure(ctx, ctx->d_ws4, (size_t)grid * 3 * 8192 * sizeof(cx<double>));
        if (rc) return rc;
        auto kern = if0_spectrum_blue4_kernel<T>;
        MPX_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(T), lds, st, yc, frames, NF, channels, power, plan.d_window, plan.d_tw,
                           plan.d_twn, plan.d_twn_r, ut, nf, (cx<double>*)ctx->d_ws4.p);
    } else if (2 * NF > 8192) {
        constexpr int T = 512;
        const size_t lds = sizeof(cx<double>) * lds_slots(8192);
        const long long grid = nf < ctx->num_cus ? nf : ctx->num_cus;   // persistent: one workgroup per CU (139 KB of LDS)
        int rc = ensure(ctx, ctx->d_ws4, (size_t)grid * 8192 * sizeof(cx<double>));
        if (rc) return rc;
        auto kern = if0_spectrum_blue2_kernel<T>;
        MPX_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(T), lds, st, yc, frames, NF, channels, power, plan.d_window, plan.d_tw,
                           plan.d_twn, plan.d_twn_r, ut, nf, (cx<double>*)ctx->d_ws4.p);
    } else {
        constexpr int L = 8192, T = 512;
        const size_t lds = sizeof(cx<double>) * lds_slots(L);
        auto kern = if0_spectrum_blue_kernel<L, T>;
        MPX_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)nf), dim3(T), lds, st, yc, frames, NF, channels, power, plan.d_window, plan.d_tw,
                           plan.d_twn, plan.d_twn_r, ut);
    }
    MPX_HIP(ctx, hipGetLastError());
    return MPX_OK;
}

template <int NF, int T>
static int if0_spectrum_launch(mpx_ctx* ctx, const double* yc, const If0Frame* frames, long long nf, int channels,
                               double power, const If0Plan& plan, double* ut, hipStream_t st) {
#ifdef MPX_DEV_KNOBS   // the two earlier spectrum kernels exist in development builds only (A/B through the environment)
    if (dev_env("MPX_IF0_STOCKHAM")) {  // profiling knob: the padded NF-point Stockham transform
        const size_t lds = sizeof(cx<double>) * lds_slots(NF);
        auto kern = if0_spectrum_kernel<NF, T>;
        if (lds > 48 * 1024)
            MPX_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)nf), dim3(T), lds, st, yc, frames, channels, power, plan.d_window, plan.d_tw,
                           plan.d_twn, ut);
    } else if (dev_env("MPX_IF0_BOTH_PARITIES")) {   // round 2: one workgroup per frame computes both parities
        const size_t lds = sizeof(cx<double>) * (NF / 2);
        auto kern = if0_spectrum_dif_kernel<NF>;
        if (lds > 48 * 1024)
            MPX_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)nf), dim3(NF / 16), lds, st, yc, frames, channels, power, plan.d_window,
                           plan.d_tw, plan.d_twn, ut);
    } else
#endif
    {
        const size_t lds = sizeof(cx<double>) * (NF / 2);
#ifdef MPX_DEV_KNOBS
        const int pf = dev_env_int("MPX_IF0_PF", 1);
        auto kern = power == 1.0 ? (pf == 2 ? if0_spectrum_split_kernel<NF, true, 2> : (pf == 1 ? if0_spectrum_split_kernel<NF, true, 1> : if0_spectrum_split_kernel<NF, true, 0>))
                                 : if0_spectrum_split_kernel<NF, false, 0>;
#else
        auto kern = power == 1.0 ? if0_spectrum_split_kernel<NF, true, 1> : if0_spectrum_split_kernel<NF, false, 0>;
#endif
        if (lds > 48 * 1024)
            MPX_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const long long groups = ((nf + 7) / 8) * 16;
        hipLaunchKernelGGL(kern, dim3((unsigned)groups), dim3(NF / 16), lds, st, yc, frames, nf, channels, power, plan.d_window,
                           plan.d_tw, plan.d_twn, plan.d_twn_r, ut);
    }
    MPX_HIP(ctx, hipGetLastError());
    return MPX_OK;
}

// Packed clips on the HOST.  chroma_frames ([F,12], may be NULL), chroma_sums ([C,12], may be NULL),
// ut ([F, 2*frame], may be NULL) are host buffers.
// dev_io: `signals` is DEVICE memory used in place, chroma_frames / chroma_sums are device buffers, the kernels are only
// enqueued on `stream` (the small host-built tables are uploaded and waited for first); otherwise host-style I/O on the
// context's stream, synchronous at return.
int if0_run_host(mpx_ctx* ctx, const float* signals, const int64_t* offsets, int num_clips, int fs,
                 const mpx_if0_params* params, double* chroma_frames, double* chroma_sums, double* ut_out, bool dev_io,
                 hipStream_t stream) {
    mpx_if0_params p = params ? *params
                              : mpx_if0_params{8192, 1.0, 70, 2.3, 0.39, 4, 1.0 / 2100.0, 1.0 / 40.0, 0.0000001, 20, 20, 20, 320, 0.66, MPX_NOTES_UNICODE};
    // 1024 / 2048 / 4096 / 8192: the tuned kernels; any other size up to 16384 samples: chirp-z (if0_spectrum_blue_kernel up to
    // 4095, if0_spectrum_blue2_kernel up to 8191, if0_spectrum_blue4_kernel above -- round 6)
    const bool blue = p.frame_size != 1024 && p.frame_size != 2048 && p.frame_size != 4096 && p.frame_size != 8192;
    if (p.frame_size < 16 || p.frame_size > 16384)
        return set_error(ctx, MPX_EUNSUPPORTED, "iterative F0: frame_size %d (supported: any size in 16 ... 16384)", p.frame_size);
    if (p.channels < 1 || p.channels > IF0_MAXCH || p.max_voices < 1 || p.max_voices > 8 || p.Q < 2 || p.Q > 32 || p.M < 2 ||
        p.M > 64 || !(p.tau_min > 0) || !(p.tau_max > p.tau_min) || fs <= 0)
        return set_error(ctx, MPX_EINVAL, "bad iterative-F0 params");
    if (p.note_names != MPX_NOTES_UNICODE && p.note_names != MPX_NOTES_ASCII)
        return set_error(ctx, MPX_EINVAL, "iterative F0: unknown note_names %d", p.note_names);
    const int NF = p.frame_size, n2 = 2 * NF;
#ifdef MPX_DEV_KNOBS   // MPX_IF0_TICKS=1: where the HOST time of a call goes (stderr, microseconds between the marks)
    static const bool ticks_on = std::getenv("MPX_IF0_TICKS") != nullptr;
    std::vector<std::pair<const char*, double>> ticks;
    auto tick = [&](const char* what) {
        if (ticks_on) ticks.push_back({what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count()});
    };
    auto ticks_out = [&]() {
        if (!ticks_on) return;
        fprintf(stderr, "mpx if0 host ticks:");
        for (size_t i = 1; i < ticks.size(); ++i) fprintf(stderr, " %s %.0f", ticks[i].first, ticks[i].second - ticks[i - 1].second);
        fprintf(stderr, " | total %.0f us\n", ticks.empty() ? 0.0 : ticks.back().second - ticks.front().second);
    };
    tick("enter");
#define IF0_TICK(w) tick(w)
#define IF0_TICKS_OUT() ticks_out()
#else
#define IF0_TICK(w) ((void)0)
#define IF0_TICKS_OUT() ((void)0)
#endif
    // periodicity.py indexes Ur up to M*K/tau_min: must stay inside the 2*frame spectrum
    if ((p.M - 1) * ((double)NF / fs) / p.tau_min + 1.5 >= n2)
        return set_error(ctx, MPX_EINVAL, "iterative F0: harmonic %d of tau_min falls outside the %d-bin spectrum (the "
                         "reference raises ValueError on the empty slice)", p.M - 1, n2);
    // The front-end output is 8 * channels bytes per sample (560 B at 70 channels): keep the workspace of one pass
    // below 32 GiB of the 288 GB (MPX_OPT_IF0_WORKSPACE_BYTES) by halving the clip list.  A clip is one serial chain per
    // channel, so a pass wants thousands of clips in flight: 4096 two-second clips take 0.38 s in passes of 256
    // (8 GiB), 0.29 s in passes of 1024.
    // Chirp-z frame sizes: a chunk is a whole number of frames AND of 64-sample tiles -- a multiple of lcm(NF, 64), the
    // smallest one of at least 65536 samples (IF0_CHUNK at most: 64 x 4095 = 262080 fits) -- and the front end writes it as ONE
    // "frame" of 2^lgp >= chunk samples per channel: [chunk][channel][2^lgp], a frame of the chunk is a piece of every row.
    long long blue_chunk = 0;
    int blue_lgp = 0;
    if (blue) {
        long long g = NF, h = 64;
        while (h) {
            const long long t = g % h;
            g = h;
            h = t;
        }
        const long long base = (long long)NF / g * 64;
        blue_chunk = base * std::max<long long>(1, (65536 + base - 1) / base);
        if (blue_chunk > IF0_CHUNK) blue_chunk = base;   // (64 x 8191 = 524 224 samples at most: the one case above IF0_CHUNK)
        while ((1LL << blue_lgp) < blue_chunk) ++blue_lgp;
    }
    // the clip list in two halves, one call each (above the workspace cap, or the 96 GiB bound further down)
    auto run_halves = [&]() -> int {
        const int mid = num_clips / 2;
        std::vector<int64_t> off2((size_t)(num_clips - mid) + 1);
        for (int i = 0; i <= num_clips - mid; ++i) off2[i] = offsets[mid + i] - offsets[mid];
        size_t frames_first = 0;
        for (int c = 0; c < mid; ++c) {
            const int64_t len = offsets[c + 1] - offsets[c];
            if (len > 0) frames_first += (size_t)((len + NF - 1) / NF);
        }
        int rc1 = if0_run_host(ctx, signals, offsets, mid, fs, &p, chroma_frames, chroma_sums, ut_out, dev_io, stream);
        if (rc1) return rc1;
        return if0_run_host(ctx, signals + (offsets[mid] - offsets[0]), off2.data(), num_clips - mid, fs, &p,
                            chroma_frames ? chroma_frames + frames_first * 12 : nullptr,
                            chroma_sums ? chroma_sums + (size_t)mid * 12 : nullptr,
                            ut_out ? ut_out + frames_first * n2 : nullptr, dev_io, stream);
    };
    {
        size_t rows = 0;
        for (int c = 0; c < num_clips; ++c) {
            const int64_t len = offsets[c + 1] - offsets[c];
            if (len < 0) return set_error(ctx, MPX_EINVAL, "offsets must be non-decreasing");
            size_t r = len <= 0 ? 0 : (size_t)((len + NF - 1) / NF) * NF;
            if (blue && r) r = (size_t)((r + blue_chunk - 1) / blue_chunk) << blue_lgp;
            rows += r;
        }
        const size_t ws_cap = ctx->if0_ws_cap;   // mpx_set_option(MPX_OPT_IF0_WORKSPACE_BYTES), default 32 GiB
        // A clip list above the cap is NOT cut while one frame of every chunk fits: it runs in time slices like a long
        // stream (below), every clip in flight -- 4096 two-second clips are 4506 waves with the leftover channels of ten
        // clips packed into one, where four passes of 1024 clips were 4 x 2048 waves with a 6-lane leftover wave per clip
        // (front end 59.6 -> see DESIGN 5.2c, ms per 4096 clips).  The list is halved for chirp-z frame sizes (no slices there)
        // and when even one frame per chunk would not fit.
        size_t est_chunks = 0;
        for (int c = 0; c < num_clips; ++c) {
            const int64_t len = offsets[c + 1] - offsets[c];
            if (len > 0) est_chunks += (size_t)((len + IF0_CHUNK - 1) / IF0_CHUNK);   // (the planner may cut finer: the cap is soft)
        }
        const bool slices_fit = !blue && est_chunks * (size_t)NF * p.channels * sizeof(double) <= ws_cap;
        if (num_clips > 1 && rows * p.channels * sizeof(double) > ws_cap && !slices_fit) return run_halves();
    }
    If0Plan plan;
    int rc = if0_plan(ctx, fs, p, plan);
    if (rc) return rc;
    double rho = 0.0;
    const long long warmup = if0_warmup(fs, p, &rho);
    // run-in of the waves of leftover channels (channels 0 .. channels % 64 - 1): their own slowest pole, floor 8192 samples
    long long warm_tail = (p.channels & 63) ? if0_warmup_range(fs, p, nullptr, 0, p.channels & 63, 8192) : warmup;
    if (!warm_tail || warm_tail > warmup) warm_tail = warmup;
    if (!warmup)
        return set_error(ctx, MPX_EINVAL, "iterative F0: the slowest pole of the filter chain has radius %.9f; chunks and time "
                         "shards start from zero state and would need more than %lld samples of run-in", rho, IF0_WARMUP_MAX);

    // chunks and frames
    std::vector<If0Chunk> chunks;
    std::vector<If0Frame> frames;
    std::vector<long long> seg(1, 0);
    long long yc_rows = 0;
    // Chunk length: a chunk is a serial chain per channel, so a long stream cut into few chunks leaves the GPU empty
    // (600 s in 262144-sample chunks: 51 chunks = 102 waves on 1024 SIMDs).  Shorter chunks fill it -- at the price of one
    // run-in of if0_warmup() samples (40960 for the default chain) per chunk; a clip shorter than that gains nothing: its
    // later chunks would re-run it from the start.
    int64_t longest = 0;
    for (int c = 0; c < num_clips; ++c) longest = std::max<int64_t>(longest, offsets[c + 1] - offsets[c]);
    auto lane_steps = [&](long long chunk) {  // samples the busiest lane of the longest clip walks through
        long long worst = 0;
        for (int64_t t0 = 0; t0 < longest; t0 += chunk)
            worst = std::max<long long>(worst, std::min<long long>(t0, warmup) + std::min<long long>(chunk, longest - t0));
        return worst;
    };
    // Two front-end kernels.  The PIPELINED one (if0_frontend_kernel) runs ONE wave per SIMD (320 registers) at 178 ns per
    // step; a launch takes ceil(waves / SIMDs) rounds of its busiest lane's steps.  The SEQUENTIAL one
    // (if0_frontend2_kernel, 165 registers, three waves per SIMD) steps a lone wave every 293 ns and n waves sharing a SIMD
    // every n x 168 ns each -- the same fp64 issue rate, but no quantisation into rounds: 1024 two-second clips are 2048
    // waves = two pipelined rounds (17.7 ms) or two sequential waves per SIMD (14.8 ms), while the hour-long stream is ONE
    // pipelined round of 923 long chunks (the sequential kernel would need twice the chunks, each with its own run-in).
    // Per kernel the chunk length that minimises the modelled time wins (the longer one on a tie: less run-in work in
    // total), then the cheaper kernel.  Times measured on MI355X (scripts/dev/if0_time.py), in units of 1/1000 pipelined step.
    // (Until round 3: "halve while there are fewer waves than SIMDs", which turned 934 chunks = 1028 waves into two
    // rounds of 98 304 steps where 467 chunks take one round of 131 072.)
    const int fe_force = dev_env_int("MPX_IF0_FE_WAVES", 0);          // development knob: 1 pipelined, 3 sequential, 0 the model's choice
    const long long simds = 4LL * ctx->num_cus;
    const int nt_ch = p.channels % 64, full_ch = p.channels / 64;
    // Candidates: every multiple of 8192 samples (a multiple of every frame size) up to IF0_CHUNK, not only the powers of
    // two: the hour of 44.1 kHz audio is 923 chunks of 172 032 samples = 1015 waves, ONE round of 212 992 steps, where
    // 262 144-sample chunks fill 65 % of the SIMDs for 303 104 steps.
    auto waves_for = [&](long long chunks_total) {
        // one wave per chunk and 64 channels, the leftover channels of up to 64 / nt chunks packed into one when that lowers
        // the number of waves the busiest SIMD takes (the same rule as below)
        long long waves = chunks_total * full_ch;
        if (nt_ch) {
            const long long pw = 64 / nt_ch;
            const long long unpacked = chunks_total * (full_ch + 1), packed = chunks_total * full_ch + (chunks_total + pw - 1) / pw;
            waves = (unpacked + simds - 1) / simds > (packed + simds - 1) / simds ? packed : unpacked;
        }
        return waves;
    };
    auto plan_chunk = [&](bool sequential, long long& best_chunk) {
        long long best_cost = -1;
        for (long long cand = IF0_CHUNK; cand >= IF0_CHUNK_MIN; cand -= 8192) {
            long long chunks_total = 0;
            for (int c = 0; c < num_clips; ++c) {
                const int64_t len = offsets[c + 1] - offsets[c];
                if (len > 0) chunks_total += (len + cand - 1) / cand;
            }
            const long long per_simd = std::max<long long>(1, (waves_for(chunks_total) + simds - 1) / simds);
            const long long unit = sequential ? std::max<long long>(1650, 944 * per_simd) : (per_simd > 1 ? 1120 : 1000) * per_simd;
            const long long cost = unit * lane_steps(cand);
            if (best_cost < 0 || cost < best_cost) {
                best_cost = cost;
                best_chunk = cand;
            }
        }
        return best_cost;
    };
    long long chunk_p = IF0_CHUNK, chunk_s = IF0_CHUNK;
    const long long cost_p = plan_chunk(false, chunk_p), cost_s = plan_chunk(true, chunk_s);
    bool fe_sequential = fe_force == 1 ? false : (fe_force == 3 ? true : cost_s < cost_p);
    long long chunk = fe_sequential ? chunk_s : chunk_p;
    if (blue) {   // one chunk length (above); the sequential kernel when the waves outnumber the SIMDs
        chunk = blue_chunk;
        long long chunks_total = 0;
        for (int c = 0; c < num_clips; ++c) {
            const int64_t len = offsets[c + 1] - offsets[c];
            if (len > 0) chunks_total += (len + chunk - 1) / chunk;
        }
        fe_sequential = fe_force == 1 ? false : (fe_force == 3 ? true : waves_for(chunks_total) > simds);
    }
    for (int c = 0; c < num_clips; ++c) {
        const int64_t len = offsets[c + 1] - offsets[c];
        if (len < 0) return set_error(ctx, MPX_EINVAL, "offsets must be non-decreasing");
        const int64_t nfr = len <= 0 ? 0 : (len + NF - 1) / NF;
        for (int64_t t0 = 0; t0 < nfr * NF; t0 += chunk) {
            If0Chunk ck;
            ck.sig_start = offsets[c] + t0;
            ck.clip_start = offsets[c];
            const int64_t want = nfr * NF - t0;            // produce whole frames (zero input beyond the clip)
            ck.len = (int)(want < chunk ? want : chunk);
            const int64_t left = len - t0;
            ck.clip_left = (int)(left > chunk + warmup ? chunk + warmup : (left > 0 ? left : 0));
            ck.warm = (int)(t0 < warmup ? t0 : warmup);
            ck.pad = 0;
            ck.yc_row0 = yc_rows;
            yc_rows += blue ? (1LL << blue_lgp) : ck.len;
            chunks.push_back(ck);
            for (int64_t fo = 0; fo < ck.len; fo += NF) {
                If0Frame fr;
                fr.yc_base = ck.yc_row0 * p.channels + (fo / NF) * (long long)p.channels * NF;   // [frame of the chunk][channel][NF]
                fr.ch_stride = NF;
                if (blue) {   // [chunk][channel][2^lgp]
                    fr.yc_base = ck.yc_row0 * p.channels + fo;
                    fr.ch_stride = 1 << blue_lgp;
                }
                fr.pad = 0;
                const int64_t fl = len - (t0 + fo);
                fr.valid = (int)(fl >= NF ? NF : (fl > 0 ? fl : 0));
                fr.clip = c;
                frames.push_back(fr);
            }
        }
        seg.push_back((long long)frames.size());
    }
    const long long nframes = (long long)frames.size(), nchunks = (long long)chunks.size();
    // leftover channels (channels % 64): consecutive chunks of equal length and run-in share a wave
    std::vector<int> tail_list;
    std::vector<If0TailGroup> tail_groups;
    const int full_groups = p.channels / 64, nt = p.channels % 64;
    if (nt) {
        // A packed wave is ~13 % slower per step (per-lane input streams), and the pass is as long as the busiest SIMD:
        // pack exactly when that lowers the number of waves the busiest SIMD has to take (800 chunks: 2 -> 1; one
        // clip: 1 -> 1, 1024 clips: 2 -> 2, where packing would only slow the leftovers down).
        const long long pw = 64 / nt;
        const long long unpacked = nchunks * (full_groups + 1), packed = nchunks * full_groups + (nchunks + pw - 1) / pw;
        const int per_wave = (unpacked + simds - 1) / simds > (packed + simds - 1) / simds ? (int)pw : 1;
        for (long long i = 0; i < nchunks; ++i) {
            if (tail_groups.empty() || tail_groups.back().count == per_wave ||
                chunks[(size_t)tail_list[(size_t)tail_groups.back().first]].len != chunks[(size_t)i].len ||
                std::min<long long>(chunks[(size_t)tail_list[(size_t)tail_groups.back().first]].warm, warm_tail) !=
                    std::min<long long>(chunks[(size_t)i].warm, warm_tail))
                tail_groups.push_back({(int)tail_list.size(), 0});
            tail_list.push_back((int)i);
            ++tail_groups.back().count;
        }
    }
    const int64_t total = offsets[num_clips];
    hipStream_t st = stream ? stream : ctx->stream;
    IF0_TICK("plan+chunks");
    if (nframes == 0) {
        if (chroma_sums && dev_io) MPX_HIP(ctx, hipMemsetAsync(chroma_sums, 0, (size_t)num_clips * 12 * sizeof(double), st));
        else if (chroma_sums) std::memset(chroma_sums, 0, (size_t)num_clips * 12 * sizeof(double));
        return MPX_OK;
    }
    // The [t][channel] hand-off buffer of the front end is the big one: 560 B per sample at 70 channels (83 GiB for an hour of
    // 44.1 kHz audio in one piece, and the first hipMalloc of that size is seconds on a device whose memory was used before).
    // Above the context's cap (MPX_OPT_IF0_WORKSPACE_BYTES; a clip LIST was halved above) the call runs in TIME SLICES: every
    // launch of the front end advances every chunk by `slice` samples (whole frames) from the filter state the launch before
    // left behind (If0Slice), the summary spectra and the period search of those frames follow, and the hand-off buffer, the
    // spectra and the search's scratch hold one slice.  The chunks -- and with them the waves in flight and the run-in work
    // -- are what they would be in one piece; the results are the same bits (tests/test_gpu_iterative_f0.py).
    long long maxlen = 0;
    for (const If0Chunk& ck : chunks) maxlen = std::max<long long>(maxlen, ck.len);
    long long slice = maxlen;
    bool sliced = false;
    // Two hand-off buffers of half the cap each (pipelined front end only): the summary spectra of slice s run NEXT TO the
    // front end of slice s + 1 -- see the slice loop below.
    const bool two_buffers = !blue && !fe_sequential && !ut_out && if0_overlap_mode(ctx) != 0;
    const size_t buf_cap = two_buffers ? ctx->if0_ws_cap / 2 : ctx->if0_ws_cap;
    if (!blue && (size_t)yc_rows * p.channels * sizeof(double) > ctx->if0_ws_cap) {   // (chirp-z frame sizes: one piece)
        long long fit = (long long)(buf_cap / ((size_t)nchunks * p.channels * sizeof(double))) / NF * NF;
        if (fit < NF) fit = NF;
        if (fit < maxlen) {
            // Among the slice lengths that fit, the one whose launches waste the least: the summary-spectrum kernel runs two
            // equally long workgroups per frame in lock-step rounds of (CUs x 16384 / frame size) -- 923 chunks x 2 frames
            // are 7.2 rounds (the eighth a fifth full: +10 %, measured 67.7 against 60.8 ms per hour of audio), x 3 frames
            // 10.8 -- and every slice costs three launch ramps (counted as a twentieth of a round).
            const long long slots = std::max<long long>(1, (long long)ctx->num_cus * 16384 / NF);
            double best = -1.0;
            for (long long cand = fit; cand >= NF; cand -= NF) {
                double cost = 0.0;
                long long nsl = 0;
                for (long long lo = 0; lo < maxlen; lo += cand, ++nsl) {
                    long long fr = 0;
                    for (const If0Chunk& ck : chunks) fr += std::max<long long>(0, std::min<long long>(lo + cand, ck.len) - lo) / NF;
                    cost += (double)((2 * fr + slots - 1) / slots);
                }
                cost += 0.05 * (double)nsl;
                if (best < 0.0 || cost < best * (1.0 - 1e-9)) {   // (the longest slice on a tie)
                    best = cost;
                    slice = cand;
                }
            }
            sliced = true;
        }
    }
    const long long nslices = (maxlen + slice - 1) / slice;
    // per slice: its frames (chunk after chunk) and the row of the call's [F, 12] output each one is
    std::vector<If0Frame> sl_frames;
    std::vector<int> sl_rows;
    std::vector<long long> sl_off(1, 0);
    std::vector<long long> chunk_frame0;   // index of a chunk's first frame in `frames`
    if (sliced) {
        long long g = 0;
        for (long long i = 0; i < nchunks; ++i) {
            chunk_frame0.push_back(g);
            chunks[(size_t)i].yc_row0 = i * slice;   // the buffer holds one slice of every chunk
            g += chunks[(size_t)i].len / NF;
        }
        sl_frames.reserve(frames.size());
        sl_rows.reserve(frames.size());
        for (long long sidx = 0; sidx < nslices; ++sidx) {
            for (long long i = 0; i < nchunks; ++i) {
                const long long lo = sidx * slice, hi = std::min<long long>(lo + slice, chunks[(size_t)i].len);
                for (long long fo = lo; fo < hi; fo += NF) {
                    If0Frame fr = frames[(size_t)(chunk_frame0[(size_t)i] + fo / NF)];
                    fr.yc_base = i * slice * p.channels + ((fo - lo) / NF) * (long long)p.channels * NF;
                    sl_frames.push_back(fr);
                    sl_rows.push_back((int)(chunk_frame0[(size_t)i] + fo / NF));
                }
            }
            sl_off.push_back((long long)sl_frames.size());
        }
    } else {
        sl_off.push_back(nframes);
    }
    long long max_slice_frames = 0;
    for (size_t i = 0; i + 1 < sl_off.size(); ++i) max_slice_frames = std::max(max_slice_frames, sl_off[i + 1] - sl_off[i]);
    const std::vector<If0Frame>& up_frames = sliced ? sl_frames : frames;
    const size_t yc_bytes = (size_t)(sliced ? nchunks * slice : yc_rows) * p.channels * sizeof(double);
    // (the estimate above counts chunks of IF0_CHUNK samples; the planner may have cut finer -- the cap is soft -- and a list
    //  that then needs more than 96 GiB even in time slices is halved like one above the cap)
    if (yc_bytes > ((size_t)96 << 30) && num_clips > 1) return run_halves();
    if (yc_bytes > ((size_t)96 << 30))
        return set_error(ctx, MPX_EUNSUPPORTED, "iterative F0: %lld chunks need %zu GiB of workspace; split the call", nchunks,
                         yc_bytes >> 30);
    const long long fe_blocks = nchunks * full_groups + (long long)tail_groups.size();
    const size_t state_bytes = sliced ? (size_t)fe_blocks * 64 * sizeof(double) * (fe_sequential ? IF0_STATE_SEQ : IF0_STATE_PIPE) : 0;
    const bool in_dev = dev_io || (total && samples_on_device(ctx, signals));   // (round 6: clips already in HBM are read in place by the host entry points too)
    if (!in_dev && (rc = ensure(ctx, ctx->d_signal, (size_t)(total ? total : 1) * sizeof(float) + 64))) return rc;
    IF0_TICK("slices");
    if ((rc = ensure(ctx, ctx->d_ws0, yc_bytes))) return rc;
    IF0_TICK("ws0");
    const bool overlap = two_buffers && sliced && nslices >= 2 && if0_overlap_ready(ctx);
    if (overlap && (rc = ensure(ctx, ctx->d_ws3, yc_bytes))) return rc;
    // the period search runs ONCE, behind the last slice, on persistent workgroups with a scratch pair each
    const bool per_big = n2 > 16384;   // spectra of more than 16 384 bins: the instantiation with the larger tables
    const char* per_key = per_big ? "if0_periodicity_big" : "if0_periodicity";
    if (int have = 0; !occupancy_lookup(ctx, per_key, &have)) {
        int occ = 0;
        if (per_big)
            MPX_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, if0_periodicity_kernel<true>, PER_T, 0));
        else
            MPX_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, if0_periodicity_kernel<false>, PER_T, 0));
        occupancy_store(ctx, per_key, occ > 0 ? occ : 1);
    }
    const long long per_grid = std::min<long long>(nframes, 2LL * ctx->num_cus * ctx->occupancy[per_key]);   // scratch slots
    (void)max_slice_frames;
    IF0_TICK("occ");
    if ((rc = ensure(ctx, ctx->d_ws1, ((size_t)nframes + 2 * (size_t)per_grid) * n2 * sizeof(double)))) return rc;   // ut | ur | ud
    IF0_TICK("ws1");
    if (sliced && (rc = ensure(ctx, ctx->d_ws2, state_bytes))) return rc;
    if ((rc = ensure(ctx, ctx->d_desc, chunks.size() * sizeof(If0Chunk) + up_frames.size() * sizeof(If0Frame) + tail_list.size() * sizeof(int) +
                                       tail_groups.size() * sizeof(If0TailGroup) + sl_rows.size() * sizeof(int) + 256))) return rc;
    if ((rc = ensure(ctx, ctx->d_frames_out, (size_t)nframes * 12 * sizeof(double)))) return rc;
    ctx->batch_layout.clear();   // d_desc / d_offsets are about to hold this call's tables (method_batch's cache)
    if ((rc = ensure(ctx, ctx->d_offsets, seg.size() * sizeof(long long)))) return rc;
    if ((rc = ensure(ctx, ctx->d_sum, (size_t)num_clips * 12 * sizeof(double)))) return rc;
    IF0_TICK("small");
    If0Chunk* d_chunks = (If0Chunk*)ctx->d_desc.p;
    If0Frame* d_frames = (If0Frame*)((char*)ctx->d_desc.p + ((chunks.size() * sizeof(If0Chunk) + 15) & ~(size_t)15));
    If0TailGroup* d_tail_groups = (If0TailGroup*)((char*)d_frames + ((up_frames.size() * sizeof(If0Frame) + 15) & ~(size_t)15));
    int* d_tail_list = (int*)((char*)d_tail_groups + ((tail_groups.size() * sizeof(If0TailGroup) + 15) & ~(size_t)15));
    int* d_rows = (int*)((char*)d_tail_list + ((tail_list.size() * sizeof(int) + 15) & ~(size_t)15));
    if (!in_dev && (rc = stage_h2d(ctx, ctx->d_signal.p, signals, (size_t)total * sizeof(float), st))) return rc;
    const float* d_in = in_dev ? signals : (const float*)ctx->d_signal.p;   // the front end reads inside the clips only
    // the tables as ONE image of their block of d_desc, through the context's pinned staging (pinned_tables, mpx_api.hip)
    {
        const size_t o_fr = (size_t)((char*)d_frames - (char*)d_chunks), o_tg = (size_t)((char*)d_tail_groups - (char*)d_chunks),
                     o_tl = (size_t)((char*)d_tail_list - (char*)d_chunks), o_rw = (size_t)((char*)d_rows - (char*)d_chunks);
        const size_t img = o_rw + (sliced ? sl_rows.size() * sizeof(int) : 0), img_al = (img + 15) & ~(size_t)15;
        const size_t seg_bytes = seg.size() * sizeof(long long);
        char* h = (char*)pinned_tables(ctx, img_al + seg_bytes);
        if (h) {
            std::memcpy(h, chunks.data(), chunks.size() * sizeof(If0Chunk));
            std::memcpy(h + o_fr, up_frames.data(), up_frames.size() * sizeof(If0Frame));
            if (nt) {
                std::memcpy(h + o_tg, tail_groups.data(), tail_groups.size() * sizeof(If0TailGroup));
                std::memcpy(h + o_tl, tail_list.data(), tail_list.size() * sizeof(int));
            }
            if (sliced) std::memcpy(h + o_rw, sl_rows.data(), sl_rows.size() * sizeof(int));
            std::memcpy(h + img_al, seg.data(), seg_bytes);
            MPX_HIP(ctx, hipMemcpyAsync(d_chunks, h, img, hipMemcpyHostToDevice, st));
            MPX_HIP(ctx, hipMemcpyAsync(ctx->d_offsets.p, h + img_al, seg_bytes, hipMemcpyHostToDevice, st));
            if (dev_io) MPX_HIP(ctx, hipStreamSynchronize(st));   // this call returns before its kernels have run: the staging is the next call's
        } else {   // no pinned memory: the vectors themselves, and the wait
            MPX_HIP(ctx, hipMemcpyAsync(d_chunks, chunks.data(), chunks.size() * sizeof(If0Chunk), hipMemcpyHostToDevice, st));
            MPX_HIP(ctx, hipMemcpyAsync(d_frames, up_frames.data(), up_frames.size() * sizeof(If0Frame), hipMemcpyHostToDevice, st));
            if (nt) {
                MPX_HIP(ctx, hipMemcpyAsync(d_tail_groups, tail_groups.data(), tail_groups.size() * sizeof(If0TailGroup), hipMemcpyHostToDevice, st));
                MPX_HIP(ctx, hipMemcpyAsync(d_tail_list, tail_list.data(), tail_list.size() * sizeof(int), hipMemcpyHostToDevice, st));
            }
            if (sliced) MPX_HIP(ctx, hipMemcpyAsync(d_rows, sl_rows.data(), sl_rows.size() * sizeof(int), hipMemcpyHostToDevice, st));
            MPX_HIP(ctx, hipMemcpyAsync(ctx->d_offsets.p, seg.data(), seg.size() * sizeof(long long), hipMemcpyHostToDevice, st));
            if (dev_io || sliced) MPX_HIP(ctx, hipStreamSynchronize(st));   // the tables above are host vectors of this call
        }
    }
    IF0_TICK("tables+sync");
    double* yc = (double*)ctx->d_ws0.p;
    double* ut_all = (double*)ctx->d_ws1.p;              // [F, n2], slice after slice
    double* ur = ut_all + (size_t)nframes * n2;          // [per_grid, n2]
    double* ud = ur + (size_t)per_grid * n2;
    int lg_nf = 0;
    while ((1 << lg_nf) < NF) ++lg_nf;
    if (blue) lg_nf = blue_lgp;   // the front end's "frame" is the chunk
    If0PerArgs a;
    a.ut = ut_all;
    a.ur = ur;
    a.ud = ud;
    a.n = n2;
    a.fs = (double)fs;
    a.K = (double)NF / (double)fs;
    a.wsize = (double)NF;
    a.max_voices = p.max_voices;
    a.note_names = p.note_names;
    a.Q = p.Q;
    a.M = p.M;
    a.tau_min = p.tau_min;
    a.tau_max = p.tau_max;
    a.tau_prec = p.tau_prec;
    a.epsilon1 = p.epsilon1;
    a.epsilon2 = p.epsilon2;
    a.gamma = p.gamma;
    a.chroma = (dev_io && chroma_frames) ? chroma_frames : (double*)ctx->d_frames_out.p;
    // Overlap (round 6, development builds, MPX_IF0_OVERLAP=1; measured and NOT adopted): the front end is one wave per SIMD on a
    // 17-stage recurrence (fp64 issue), the summary spectra are bound by LDS round trips and barriers -- each leaves idle what
    // the other needs.  With two hand-off buffers the spectra of slice s go to a second stream and run beside the front end of
    // slice s + 1 (which depends on the front end of slice s alone: the carried filter state); the front end of slice s + 2
    // waits for the spectra of slice s to let go of its buffer.  For the two to share a CU the front end's tile must be 32
    // samples wide (if0_frontend_kernel<true, 32>: 17 KB of LDS per wave; four of them and ONE 64 KB spectrum workgroup fit in
    // 160 KB, 216 + 2 x 128 registers per SIMD).  Same bits (scripts/dev/if0_hour_caps.py 12 24o), and the hour takes 100-102 ms
    // instead of 91-92 (profiles/r6/if0_overlap_ab.txt): the 32-sample tile alone costs the front end 42.8 -> 66.4 ms (256-byte
    // pieces of 64 rows per flush: the HBM write pattern the 64-sample tile was built to avoid), and the spectra at ONE
    // workgroup per CU beside it take 75 ms instead of 42.6 -- the slower of the two sets the pace.  s_setprio on the front
    // end's waves changes nothing (64.6 / 74.2 ms without it).
    hipStream_t sp_st = overlap ? ctx->if0_sp_stream : st;
    for (long long sidx = 0; sidx < nslices; ++sidx) {
        If0Slice sl;
        sl.t0 = (int)(sidx * slice);
        sl.t1 = sliced ? (int)((sidx + 1) * slice) : 0x7fffffff;
        sl.state = sliced ? (double*)ctx->d_ws2.p : nullptr;
        const long long nf_s = sl_off[(size_t)sidx + 1] - sl_off[(size_t)sidx];
        const int bi = overlap ? (int)(sidx & 1) : 0;
        double* yc_s = bi ? (double*)ctx->d_ws3.p : yc;
        if (overlap && sidx >= 2) MPX_HIP(ctx, hipStreamWaitEvent(st, ctx->if0_ev_sp[bi], 0));   // the buffer's last reader
        prof_mark(ctx, st, "if0_frontend_kernel");
        const bool tw32 = sliced && !fe_sequential && (overlap || dev_env_on("MPX_IF0_TW32"));
        if (fe_sequential) {
            auto fe_kernel = sliced ? if0_frontend2_kernel<true> : if0_frontend2_kernel<false>;
            hipLaunchKernelGGL(fe_kernel, dim3((unsigned)fe_blocks), dim3(64), 0, st, d_in, d_chunks, nchunks, p.channels,
                               plan.d_coefs, plan.wf, yc_s, d_tail_list, d_tail_groups, (int)tail_groups.size(), lg_nf, sl, (int)warm_tail,
                               blue ? plan.d_one : plan.d_window, blue ? 0 : NF - 1);   // the window goes on in the front end's tile flush
        } else {
            auto fe_kernel = sliced ? if0_frontend_kernel<true> : if0_frontend_kernel<false>;
            if constexpr (DEV_KNOBS != 0)
                if (tw32) fe_kernel = if0_frontend_kernel<true, 32>;
            hipLaunchKernelGGL(fe_kernel, dim3((unsigned)fe_blocks), dim3(64), 0, st, d_in, d_chunks, nchunks, p.channels,
                               plan.d_coefs, plan.wf, yc_s, d_tail_list, d_tail_groups, (int)tail_groups.size(), lg_nf, sl, (int)warm_tail,
                               blue ? plan.d_one : plan.d_window, blue ? 0 : NF - 1, overlap ? dev_env_int("MPX_IF0_FE_PRIO", 1) : 0);
        }
        MPX_HIP(ctx, hipGetLastError());
        if (overlap) {
            prof_mark(ctx, st, nullptr);
            MPX_HIP(ctx, hipEventRecord(ctx->if0_ev_fe[bi], st));
            MPX_HIP(ctx, hipStreamWaitEvent(sp_st, ctx->if0_ev_fe[bi], 0));
        }
        if (nf_s == 0) {
            if (overlap) MPX_HIP(ctx, hipEventRecord(ctx->if0_ev_sp[bi], sp_st));
            continue;
        }
        const If0Frame* d_fr = d_frames + sl_off[(size_t)sidx];
        double* ut = ut_all + (size_t)sl_off[(size_t)sidx] * n2;
        prof_mark(ctx, sp_st, "if0_spectrum_kernel");
        if (blue) rc = if0_spectrum_blue_launch(ctx, yc_s, d_fr, nf_s, NF, p.channels, p.power, plan, ut, sp_st);
        else if (NF == 1024) rc = if0_spectrum_launch<1024, 64>(ctx, yc_s, d_fr, nf_s, p.channels, p.power, plan, ut, sp_st);
        else if (NF == 2048) rc = if0_spectrum_launch<2048, 128>(ctx, yc_s, d_fr, nf_s, p.channels, p.power, plan, ut, sp_st);
        else if (NF == 4096) rc = if0_spectrum_launch<4096, 256>(ctx, yc_s, d_fr, nf_s, p.channels, p.power, plan, ut, sp_st);
        else rc = if0_spectrum_launch<8192, 512>(ctx, yc_s, d_fr, nf_s, p.channels, p.power, plan, ut, sp_st);
        if (rc) return rc;
        prof_mark(ctx, sp_st, nullptr);
        if (overlap) MPX_HIP(ctx, hipEventRecord(ctx->if0_ev_sp[bi], sp_st));
        if (ut_out) {   // (never with two buffers)
            if (!sliced) {
                MPX_HIP(ctx, hipMemcpyAsync(ut_out, ut, (size_t)nframes * n2 * sizeof(double), hipMemcpyDeviceToHost, st));
            } else {   // a run of local rows that belongs to one chunk is a run of the call's rows
                long long j = sl_off[(size_t)sidx];
                while (j < sl_off[(size_t)sidx + 1]) {
                    long long e = j + 1;
                    while (e < sl_off[(size_t)sidx + 1] && sl_rows[(size_t)e] == sl_rows[(size_t)e - 1] + 1) ++e;
                    MPX_HIP(ctx, hipMemcpyAsync(ut_out + (size_t)sl_rows[(size_t)j] * n2, ut + (size_t)(j - sl_off[(size_t)sidx]) * n2,
                                                (size_t)(e - j) * n2 * sizeof(double), hipMemcpyDeviceToHost, st));
                    j = e;
                }
            }
        }
    }
    if (overlap) {   // the period search reads every slice's spectra
        MPX_HIP(ctx, hipStreamWaitEvent(st, ctx->if0_ev_sp[(nslices - 1) & 1], 0));
        if (nslices >= 2) MPX_HIP(ctx, hipStreamWaitEvent(st, ctx->if0_ev_sp[(nslices - 2) & 1], 0));
    }
    IF0_TICK("slice launches");
    a.out_row = sliced ? d_rows : nullptr;
    a.num_frames = nframes;
    if (ctx->d_queue.bytes < (size_t)per_grid * sizeof(unsigned)) {
        if ((rc = ensure(ctx, ctx->d_queue, (size_t)per_grid * sizeof(unsigned) + 4096))) return rc;
        MPX_HIP(ctx, hipMemsetAsync(ctx->d_queue.p, 0, ctx->d_queue.bytes, st));   // every launch leaves its slots free
    }
    a.slot_busy = (unsigned*)ctx->d_queue.p;
    a.num_slots = (int)per_grid;
    // (Measured and rejected at the end of round 3: the period search of one group of 2048 frames on a second stream next to
    //  the summary spectra of the next group -- it fits beside them on every CU, but the whole-hour call went from 126.5 to
    //  135 ms: its re-reads of the spectrum rows (6.9 x their bytes, through L2) slow the LDS/L2-bound spectra down by more
    //  than its own 25 ms.)
    prof_mark(ctx, st, "if0_periodicity_kernel");
    if (per_big)
        hipLaunchKernelGGL(if0_periodicity_kernel<true>, dim3((unsigned)nframes), dim3(PER_T), 0, st, a);
    else
        hipLaunchKernelGGL(if0_periodicity_kernel<false>, dim3((unsigned)nframes), dim3(PER_T), 0, st, a);
    prof_mark(ctx, st, nullptr);
    MPX_HIP(ctx, hipGetLastError());
    IF0_TICK("period launch");
    if (dev_io) {
        IF0_TICKS_OUT();
        if (chroma_sums) return segment_sum(ctx, a.chroma, (const long long*)ctx->d_offsets.p, num_clips, nframes, chroma_sums, st);
        return MPX_OK;
    }
    if (chroma_frames)
        MPX_HIP(ctx, hipMemcpyAsync(chroma_frames, ctx->d_frames_out.p, (size_t)nframes * 12 * sizeof(double), hipMemcpyDeviceToHost, st));
    if (chroma_sums) {
        if ((rc = segment_sum(ctx, (const double*)ctx->d_frames_out.p, (const long long*)ctx->d_offsets.p, num_clips, nframes,
                              (double*)ctx->d_sum.p, st)))
            return rc;
        MPX_HIP(ctx, hipMemcpyAsync(chroma_sums, ctx->d_sum.p, (size_t)num_clips * 12 * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    MPX_HIP(ctx, hipStreamSynchronize(st));
    IF0_TICK("sync");
    IF0_TICKS_OUT();
    return MPX_OK;
}

// Debug tap / drop-in for IterativeF0PeriodicityAnalysis.compute (periodicity.py:48-163): the period search alone on summary
// spectra the CALLER hands in (host memory, [F, n2] doubles, n2 = 2 x window size), per-frame chroma out.
int if0_periodicity_host(mpx_ctx* ctx, const double* spectra, long long nframes, int n2, int fs, const mpx_if0_params* params,
                         double* chroma_frames, double* saliences, double* periods) {
    mpx_if0_params p = params ? *params
                              : mpx_if0_params{8192, 1.0, 70, 2.3, 0.39, 4, 1.0 / 2100.0, 1.0 / 40.0, 0.0000001, 20, 20, 20, 320, 0.66, MPX_NOTES_UNICODE};
    if (fs <= 0 || nframes < 0 || n2 < 32 || (n2 & 1) || n2 != 2 * p.frame_size || !chroma_frames || (nframes && !spectra))
        return set_error(ctx, MPX_EINVAL, "iterative F0 periodicity: need [F, 2 x frame_size] spectra (got n2 = %d, frame_size = %d)", n2, p.frame_size);
    // (the same bounds as the whole method, if0_run_host)
    if (p.max_voices < 1 || p.max_voices > 8 || p.Q < 2 || p.Q > 32 || p.M < 2 || p.M > 64 || !(p.tau_min > 0) || !(p.tau_max > p.tau_min))
        return set_error(ctx, MPX_EINVAL, "bad iterative-F0 params");
    if (p.note_names != MPX_NOTES_UNICODE && p.note_names != MPX_NOTES_ASCII)
        return set_error(ctx, MPX_EINVAL, "iterative F0: unknown note_names %d", p.note_names);
    if (p.frame_size < 16 || p.frame_size > 16384)
        return set_error(ctx, MPX_EUNSUPPORTED, "iterative F0: frame_size %d (supported: any size in 16 ... 16384)", p.frame_size);
    if ((p.M - 1) * ((double)p.frame_size / fs) / p.tau_min + 1.5 >= n2)
        return set_error(ctx, MPX_EINVAL, "iterative F0: harmonic %d of tau_min falls outside the %d-bin spectrum (the "
                         "reference raises ValueError on the empty slice)", p.M - 1, n2);
    if (nframes == 0) return MPX_OK;
    hipStream_t st = ctx->stream;
    int rc;
    const bool per_big = n2 > 16384;
    const char* per_key = per_big ? "if0_periodicity_big" : "if0_periodicity";
    if (int have = 0; !occupancy_lookup(ctx, per_key, &have)) {
        int occ = 0;
        if (per_big)
            MPX_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, if0_periodicity_kernel<true>, PER_T, 0));
        else
            MPX_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, if0_periodicity_kernel<false>, PER_T, 0));
        occupancy_store(ctx, per_key, occ > 0 ? occ : 1);
    }
    const long long per_grid = std::min<long long>(nframes, 2LL * ctx->num_cus * ctx->occupancy[per_key]);
    if ((rc = ensure(ctx, ctx->d_ws1, ((size_t)nframes + 2 * (size_t)per_grid) * n2 * sizeof(double)))) return rc;   // ut | ur | ud
    if ((rc = ensure(ctx, ctx->d_frames_out, (size_t)nframes * (12 + 16) * sizeof(double)))) return rc;   // chroma rows | voices rows
    double* ut_all = (double*)ctx->d_ws1.p;
    MPX_HIP(ctx, hipMemcpyAsync(ut_all, spectra, (size_t)nframes * n2 * sizeof(double), hipMemcpyHostToDevice, st));
    If0PerArgs a;
    a.ut = ut_all;
    a.ur = ut_all + (size_t)nframes * n2;
    a.ud = a.ur + (size_t)per_grid * n2;
    a.n = n2;
    a.fs = (double)fs;
    a.K = (double)p.frame_size / (double)fs;
    a.wsize = (double)p.frame_size;
    a.max_voices = p.max_voices;
    a.note_names = p.note_names;
    a.Q = p.Q;
    a.M = p.M;
    a.tau_min = p.tau_min;
    a.tau_max = p.tau_max;
    a.tau_prec = p.tau_prec;
    a.epsilon1 = p.epsilon1;
    a.epsilon2 = p.epsilon2;
    a.gamma = p.gamma;
    a.chroma = (double*)ctx->d_frames_out.p;
    a.out_row = nullptr;
    a.voices = (saliences || periods) ? (double*)ctx->d_frames_out.p + (size_t)nframes * 12 : nullptr;
    a.num_frames = nframes;
    if (ctx->d_queue.bytes < (size_t)per_grid * sizeof(unsigned)) {
        if ((rc = ensure(ctx, ctx->d_queue, (size_t)per_grid * sizeof(unsigned) + 4096))) return rc;
        MPX_HIP(ctx, hipMemsetAsync(ctx->d_queue.p, 0, ctx->d_queue.bytes, st));   // every launch leaves its slots free
    }
    a.slot_busy = (unsigned*)ctx->d_queue.p;
    a.num_slots = (int)per_grid;
    prof_mark(ctx, st, "if0_periodicity_kernel");
    if (per_big)
        hipLaunchKernelGGL(if0_periodicity_kernel<true>, dim3((unsigned)nframes), dim3(PER_T), 0, st, a);
    else
        hipLaunchKernelGGL(if0_periodicity_kernel<false>, dim3((unsigned)nframes), dim3(PER_T), 0, st, a);
    prof_mark(ctx, st, nullptr);
    MPX_HIP(ctx, hipGetLastError());
    MPX_HIP(ctx, hipMemcpyAsync(chroma_frames, ctx->d_frames_out.p, (size_t)nframes * 12 * sizeof(double), hipMemcpyDeviceToHost, st));
    std::vector<double> hv;
    if (a.voices) {
        hv.resize((size_t)nframes * 16);
        MPX_HIP(ctx, hipMemcpyAsync(hv.data(), a.voices, hv.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    MPX_HIP(ctx, hipStreamSynchronize(st));
    for (long long f = 0; a.voices && f < nframes; ++f)
        for (int i = 0; i < p.max_voices; ++i) {
            if (saliences) saliences[f * p.max_voices + i] = hv[(size_t)f * 16 + i];
            if (periods) periods[f * p.max_voices + i] = hv[(size_t)f * 16 + 8 + i];
        }
    return MPX_OK;
}

int if0_warmup_samples(mpx_ctx* ctx, int fs, const mpx_if0_params* params, long long* samples, double* rho) {
    mpx_if0_params p = params ? *params
                              : mpx_if0_params{8192, 1.0, 70, 2.3, 0.39, 4, 1.0 / 2100.0, 1.0 / 40.0, 0.0000001, 20, 20, 20, 320, 0.66, MPX_NOTES_UNICODE};
    if (p.channels < 1 || p.channels > IF0_MAXCH || fs <= 0) return set_error(ctx, MPX_EINVAL, "bad iterative-F0 params");
    double r = 0.0;
    const long long w = if0_warmup(fs, p, &r);
    if (rho) *rho = r;
    if (!w)
        return set_error(ctx, MPX_EINVAL, "iterative F0: slowest pole radius %.9f needs more than %lld samples of run-in", r, IF0_WARMUP_MAX);
    *samples = w;
    return MPX_OK;
}

}  // namespace mpx
