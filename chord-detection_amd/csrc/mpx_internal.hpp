// Host-side context, plan caches and small helpers shared by the kernel files.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <ctime>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/mpx.h"

namespace mpx {

// Where a frame's samples live: start index into the signal buffer and how many
// of its `frame` samples exist (the rest is the zero padding of dsp/frame.py:11-12).
struct FrameDesc {
    long long start;
    int valid;
    int clip;
};

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

// Harmonic-energy plan: everything that depends only on (fs, N, params, dtype).
struct HePlan {
    void* window = nullptr;  // cx[N]     (cos, sin)(2 pi s/(N-1)): symmetric Hamming angles (harmonic_energy.py:42)
    void* tw = nullptr;      // cx[M]     W_M^j, M = N/2
    void* twn = nullptr;     // cx[M+1]   W_N^k (real-FFT split)
    int* wk0 = nullptr;      // [nwin] window start (index into `bins`)
    int* wk1 = nullptr;      // [nwin] window end (exclusive, harmonic_energy.py:58)
    std::vector<int> h_bins;  // [nb] the bins some window looks at, ascending (host)
    std::vector<int> h_k0, h_k1;  // [nwin] window bin ranges [k0, k1) (host)
    mutable unsigned* slots = nullptr;  // [nb] LDS slots of Z[k], Z[M-k] for the engine of this frame size (he_launch)
    mutable void* whalf = nullptr;      // he_wave_kernel (N = 4096, fp64): hamming_sym(N)[0 .. N/2), double
    mutable unsigned* wslots = nullptr; //   [nb][2] positions of ZA[k'], ZA[-k'] | ZB[k'], ZB[-k'] in its bin-ordered LDS copy
    mutable int quad_tail = -1;         //   the reference's window shape (one window per lane, a note per quad)?  -1: not decided
    void* twnb = nullptr;    // cx[nb] W_N^k at those bins (so that the load does not wait for bins[i])
    int nb = 0;
    void* ww = nullptr;      // Real[nwin] 1/harmonic
    int nwin = 0;            // 12 * num_octave * num_harmonic
    int wins_per_note = 0;
    int num_harmonic = 0;
    int kmin = 0, kmax = 0;  // bins needed: [kmin, kmax)
    bool wrapped = false;    // some window starts below bin 0 and wraps to the top of the spectrum (he_windows, mpx_he.hip)
};

}  // namespace mpx

struct mpx_ctx {
    int device = 0;
    int flags = 0;
    int num_cus = 256;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;        // host batches: the copy of piece k+1 runs next to the kernels of piece k
    hipEvent_t copy_ev[8] = {};
    bool copy_ready = false;                  // copy_stream and all of copy_ev exist
    int copy_pieces = 4;                      // pieces of a large host batch (1: no overlap); fixed at mpx_create
    hipStream_t side_stream = nullptr;        // ESACF, small batches: coopfit_live_kernel runs here, next to the lane kernel (mpx_esacf.hip)
    hipEvent_t side_ev[2] = {};
    bool side_ready = false;
    unsigned live_epoch = 0;                  // tag of the parked records of the current batch (never 0)
    void* h_results = nullptr;                // pinned staging for result copies of batches ([clips, 12] doubles): d2h_results
    size_t h_results_bytes = 0;
    void* h_tables = nullptr;                 // pinned staging for host-built tables a call uploads (pinned_tables, mpx_api.hip)
    size_t h_tables_bytes = 0;
    size_t if0_ws_cap = (size_t)32 << 30;     // MPX_OPT_IF0_WORKSPACE_BYTES
    hipStream_t if0_sp_stream = nullptr;      // development builds, MPX_IF0_OVERLAP=1: the summary spectra's stream (if0_run_host)
    hipEvent_t if0_ev_fe[2] = {}, if0_ev_sp[2] = {};
    bool if0_overlap_made = false;
    int he_kernel = 0;                        // MPX_OPT_HE_KERNEL
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;
    std::map<std::tuple<int, int, int, int, int>, mpx::HePlan> he_plans;
    std::map<std::string, std::vector<void*>> misc_plans;
    // grow-only device workspaces
    mpx::DevBuf d_pcm, d_pcm_f32;   // PCM_16 entry points: the int16 samples as copied, and (methods 3 / 4) their float32 conversion
    mpx::DevBuf d_signal, d_frames_out, d_partials, d_sum, d_desc, d_offsets, d_ws0, d_ws1, d_ws2, d_ws3, d_ws4, d_counter, d_queue;   // (d_ws4: sacf_huge_kernel accumulators; d_counter: he_kernel's tickets, left at zero; d_queue: scratch-slot flags, all free between launches)
    std::map<std::string, std::vector<unsigned char>> host_blobs;  // host copies of plan records, per context
    std::map<std::string, int> occupancy;      // cached hipOccupancyMaxActiveBlocksPerMultiprocessor answers
    std::map<int, std::vector<double>> remez;  // user-registered warped-FIR taps per sample rate
    // method_batch: the clip layout (offsets, frame, hop) whose frame descriptors and segment table d_desc / d_offsets hold --
    // a corpus driver calls with the same layout chunk after chunk; empty: they hold something else (Iterative-F0, Prime-multiF0)
    std::vector<int64_t> batch_layout;
    int64_t batch_layout_frames = 0;
    std::vector<void*> owned;  // plan tables, freed in mpx_destroy
    std::vector<void*> retired, retired_host;   // blocks a workspace / the pinned result staging has outgrown (ensure, mpx_api.hip)
    size_t retired_bytes = 0;
    size_t retired_cap = (size_t)8 << 30;       // ... freed together once they hold more than this (mpx_create: an eighth of the device's memory)
    // per-kernel timing (mpx_profile_begin / mpx_profile_end): an event in front of every launch while enabled
    struct ProfMark {
        const char* name;  // kernel launched right after the event; nullptr closes the previous region
        hipEvent_t ev;
    };
    long long fit_stats[3] = {0, 0, 0};   // gaussian fits, MINPACK function evaluations, parked fits since mpx_profile_begin
    std::atomic<unsigned> launches{0};   // mpx_launch_count
    bool prof_on = false;
    std::vector<ProfMark> prof_marks;
    std::vector<hipEvent_t> prof_pool;
};

namespace mpx {

// Development switches -- A/B, ablation and end-game tuning aids of csrc/, some of which change results -- exist only in
// builds made with -DMPX_DEV_KNOBS (`make dev` -> libmpx_hip_dev.so, loaded by the tools under tests/tools and scripts/
// through MPX_LIB_PATH).  The release library never looks at the environment for them: dev_env() is a constant there.
// (MPX_DETERMINISTIC is not one of them: it selects MPX_FLAG_DETERMINISTIC, the same bits computed the slower way.)
#ifdef MPX_DEV_KNOBS
inline const char* dev_env(const char* name) { return getenv(name); }
constexpr int DEV_KNOBS = 1;
#else
inline const char* dev_env(const char*) { return nullptr; }
constexpr int DEV_KNOBS = 0;
#endif
inline int dev_env_int(const char* name, int dflt) {
    const char* v = dev_env(name);
    return v ? atoi(v) : dflt;
}
inline bool dev_env_on(const char* name) { return dev_env_int(name, 0) != 0; }
// MPX_TICKS=1 (development builds): where the HOST time of a call goes -- microseconds since the calling thread's previous tick
#ifdef MPX_DEV_KNOBS
inline void dev_tick(const void* ctx, const char* what) {
    static const bool on = getenv("MPX_TICKS") != nullptr;
    if (!on) return;
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    const double now = ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
    static thread_local double prev = 0.0;
    fprintf(stderr, "mpx tick %p %-28s +%.0f us (t = %.3f ms)\n", ctx, what, prev ? now - prev : 0.0, std::fmod(now * 1e-3, 1e6));
    prev = now;
}
#else
inline void dev_tick(const void*, const char*) {}
#endif

int set_error(mpx_ctx* ctx, int code, const char* fmt, ...);
int ensure(mpx_ctx* ctx, DevBuf& b, size_t bytes);
bool samples_on_device(const mpx_ctx* ctx, const void* p);
void* pinned_tables(mpx_ctx* ctx, size_t bytes);
bool occupancy_lookup(mpx_ctx* ctx, const std::string& key, int* v);   // the context's cache, then the process's (per device)
void occupancy_store(mpx_ctx* ctx, const std::string& key, int v);   // >= bytes of pinned host memory owned by the context, or nullptr
void* upload(mpx_ctx* ctx, const void* host, size_t bytes);  // nullptr on failure (error set)
// Samples (host or device memory) into device memory, enqueued on `st` (see mpx_api.hip for the measured rates).
int stage_h2d(mpx_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t st);
// Results of a batch back to the caller's (pageable) buffer and the stream synchronised: above 8 KB through a pinned buffer of the
// context's -- the runtime's own path for a pageable destination stages and waits per piece -- then one host memcpy.
int d2h_results_sync(mpx_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes, hipStream_t st);
// While profiling is on: record an event on `st`; the time to the next mark is booked on `name` (nullptr: on nothing).
void prof_mark_slow(mpx_ctx* ctx, hipStream_t st, const char* name);
inline void prof_mark(mpx_ctx* ctx, hipStream_t st, const char* name) {
    if (name) ctx->launches.fetch_add(1u, std::memory_order_relaxed);   // mpx_launch_count: a kernel (group) is about to be enqueued
    if (ctx->prof_on) prof_mark_slow(ctx, st, name);
}

#define MPX_HIP(ctx, call)                                                                   \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess)                                                                \
            return mpx::set_error((ctx), MPX_EHIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// he
int he_argmax_run(mpx_ctx* ctx, const float* d_signal, int64_t n, int64_t num_frames, int fs, const mpx_he_params* params,
                  int frame, int hop, int* d_argmax, int* h_bounds, hipStream_t stream);
int he_run(mpx_ctx* ctx, const float* d_signal, int64_t n, const FrameDesc* d_desc, int64_t num_frames,
           int fs, const mpx_he_params* params, int frame, int hop, double* d_chroma_frames,
           double* d_chroma_sum, hipStream_t stream);
// segmented sum of per-frame chroma: out[s] = sum_{f in [seg[s], seg[s+1])} frames[f]
int segment_sum(mpx_ctx* ctx, const double* d_frames, const long long* d_seg, int num_seg,
                int64_t num_frames, double* d_out, hipStream_t stream);
// esacf
int esacf_run(mpx_ctx* ctx, const float* d_signal, int64_t n, const FrameDesc* d_desc,
              int64_t num_frames, int fs, const mpx_esacf_params* params, int frame, int hop,
              double* d_chroma_frames, int stage, double* d_stage_out, hipStream_t stream);

int if0_run_host(mpx_ctx* ctx, const float* signals, const int64_t* offsets, int num_clips, int fs,
                 const mpx_if0_params* params, double* chroma_frames, double* chroma_sums, double* ut_out,
                 bool dev_io = false, hipStream_t stream = nullptr);
int if0_periodicity_host(mpx_ctx* ctx, const double* spectra, long long nframes, int n2, int fs, const mpx_if0_params* params,
                         double* chroma_frames, double* saliences = nullptr, double* periods = nullptr);
int if0_warmup_samples(mpx_ctx* ctx, int fs, const mpx_if0_params* params, long long* samples, double* rho);
int prime_run_host(mpx_ctx* ctx, const float* signals, const int64_t* offsets, int num_clips, int fs,
                   const mpx_prime_params* params, double* chroma_sums, bool dev_io = false, hipStream_t stream = nullptr);

}  // namespace mpx
