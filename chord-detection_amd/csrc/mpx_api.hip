// C ABI (include/mpx.h): context, workspaces, host<->device staging.
#include <cmath>
#include <cstdarg>
#include <chrono>
#include <mutex>
#include <cstring>

#include "mpx_internal.hpp"
#include "mpx_lm.hpp"
#include "mpx_pow067.hpp"

static thread_local std::string g_create_error;

namespace mpx {

int set_error(mpx_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->err = buf;
    else
        g_create_error = buf;
    return code;
}

// Blocks a workspace has outgrown.  hipFree / hipHostFree wait for the whole DEVICE, other contexts' streams included, and a
// corpus driver runs the methods on three contexts at once: a context that outgrows a workspace next to another context's
// running kernels waits for all of them.  Measured with MPX_TICKS=1 (development library, profiles/r6/corpus_cold_ticks.txt): the
// ESACF call of the first 4096-clip group of a process sat 24.7 ms between its plan lookup and its first launch -- the length of
// the Prime-multiF0 kernel running beside it -- and Iterative-F0, which waits for that launch (corpus._start_side), with it.
// So an outgrown block is RETIRED: kept until the context is destroyed, until an allocation fails, or until more than
// ctx->retired_cap bytes are held that way (an eighth of the device's memory, at least 8 GiB: the 12 -> 83 GiB hand-off buffer
// of a long stream retires 12 GiB) -- then everything retired is freed at once, with the wait.  (A first cap of 1 GiB was below
// what the 1024-clip workspaces of that driver add up to and changed nothing: profiles/r6/corpus_growth_ab.txt.)
static void release_retired(mpx_ctx* ctx) {
    if (ctx->retired.empty() && ctx->retired_host.empty()) return;
    hipStreamSynchronize(ctx->stream);   // work queued on our stream may still use them
    for (void* p : ctx->retired) (void)hipFree(p);
    for (void* p : ctx->retired_host) (void)hipHostFree(p);
    ctx->retired.clear();
    ctx->retired_host.clear();
    ctx->retired_bytes = 0;
}
void release_retired_blocks(mpx_ctx* ctx) { release_retired(ctx); }   // mpx_destroy

// hipOccupancyMaxActiveBlocksPerMultiprocessor costs 1.3 ms the first time a CONTEXT asks (MPX_IF0_TICKS, "occ"), whatever the
// process has asked before: the answers are kept per (device, kernel) for the whole process as well.
static std::mutex g_occ_mutex;
static std::map<std::pair<int, std::string>, int> g_occ;
bool occupancy_lookup(mpx_ctx* ctx, const std::string& key, int* v) {
    auto it = ctx->occupancy.find(key);
    if (it != ctx->occupancy.end()) {
        *v = it->second;
        return true;
    }
    std::lock_guard<std::mutex> lock(g_occ_mutex);
    auto g = g_occ.find({ctx->device, key});
    if (g == g_occ.end()) return false;
    ctx->occupancy[key] = *v = g->second;
    return true;
}
void occupancy_store(mpx_ctx* ctx, const std::string& key, int v) {
    ctx->occupancy[key] = v;
    std::lock_guard<std::mutex> lock(g_occ_mutex);
    g_occ[{ctx->device, key}] = v;
}

// Is p memory of the context's OWN device (hipMalloc'ed, a torch tensor's storage ...)?  The host entry points read such samples
// in place; another device's memory goes through the staging copy like host memory (hipMemcpyDefault finds its way).
bool samples_on_device(const mpx_ctx* ctx, const void* p) {
    hipPointerAttribute_t attr;
    if (p && hipPointerGetAttributes(&attr, p) == hipSuccess) return attr.type == hipMemoryTypeDevice && attr.device == ctx->device;
    (void)hipGetLastError();   // plain pageable memory is "invalid value" to this query, not an error
    return false;
}

// Pinned staging for tables a call builds on the host and uploads: a copy from PAGEABLE memory keeps the calling thread until it
// has run, one copy after the other, and next to other contexts' kernels each waits for room on the GPU -- Iterative-F0's six
// table uploads took 5-6 ms in the corpus driver where they take 75 us alone (MPX_IF0_TICKS, profiles/r6/corpus_cold_ticks.txt).
// From pinned memory they are queued like the kernels behind them.  One buffer per context, reused by the next call: a caller
// whose call RETURNS before its copies have run (the *_dev entry points) synchronises first.
void* pinned_tables(mpx_ctx* ctx, size_t bytes) {
    if (ctx->h_tables_bytes >= bytes) return ctx->h_tables;
    if (ctx->h_tables) ctx->retired_host.push_back(ctx->h_tables);   // (hipHostFree waits for the device: see ensure)
    ctx->h_tables = nullptr;
    ctx->h_tables_bytes = 0;
    void* hp = nullptr;
    const size_t want = bytes + bytes / 4 + 4096;
    if (hipHostMalloc(&hp, want, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    ctx->h_tables = hp;
    ctx->h_tables_bytes = want;
    return hp;
}

int ensure(mpx_ctx* ctx, DevBuf& b, size_t bytes) {
    if (bytes <= b.bytes) return MPX_OK;
    if (b.p) {
        ctx->retired.push_back(b.p);
        ctx->retired_bytes += b.bytes;
        b.p = nullptr;
        b.bytes = 0;
        if (ctx->retired_bytes > ctx->retired_cap || dev_env_on("MPX_ENSURE_FREE")) release_retired(ctx);   // (the knob: rounds 1-5, for the A/B)
    }
    size_t want = bytes + bytes / 4 + 256;
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(&b.p, want);
    if (dev_env_on("MPX_ENSURE_TRACE"))
        fprintf(stderr, "mpx ensure: ctx %p hipMalloc(%zu) took %.0f us\n", (void*)ctx, want,
                std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    if (e != hipSuccess && !ctx->retired.empty()) {   // what the retired blocks hold may be what is missing
        (void)hipGetLastError();
        release_retired(ctx);
        e = hipMalloc(&b.p, want);
    }
    if (e != hipSuccess) {
        b.p = nullptr;
        return set_error(ctx, MPX_ENOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    b.bytes = want;
    return MPX_OK;
}

void* upload(mpx_ctx* ctx, const void* host, size_t bytes) {
    void* d = nullptr;
    if (hipMalloc(&d, bytes ? bytes : 16) != hipSuccess) {
        set_error(ctx, MPX_ENOMEM, "hipMalloc(%zu) for a plan table failed", bytes);
        return nullptr;
    }
    if (bytes && hipMemcpy(d, host, bytes, hipMemcpyHostToDevice) != hipSuccess) {
        set_error(ctx, MPX_EHIP, "hipMemcpy of a plan table failed");
        hipFree(d);
        return nullptr;
    }
    ctx->owned.push_back(d);
    return d;
}

// Samples into device memory, enqueued on `st`.  One hipMemcpyAsync whatever the source: measured on the MI355X box
// (scripts/h2d_probe.py, profiles/r2/h2d_probe.json) the runtime moves PAGEABLE host memory at 45 GB/s for the 33.5 MB
// headline signal and 55 GB/s for a 1.4 GB clip batch (it pins the caller's pages and DMAs from them), pinned memory
// at 49 GB/s -- a staging ring of our own (four host threads copying 2 MiB pieces into pinned buffers, two each) reached
// 33 and 39 GB/s and was removed again: a memcpy per byte costs more than the page pinning it avoids.
int stage_h2d(mpx_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t st) {
    if (!bytes) return MPX_OK;
    MPX_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, st));
    return MPX_OK;
}

// The context's pinned staging for batch results ([clips, 12] doubles), at least `bytes` long; nullptr without pinned memory.
static void* pinned_results(mpx_ctx* ctx, size_t bytes) {
    if (ctx->h_results_bytes >= bytes) return ctx->h_results;
    if (ctx->h_results) ctx->retired_host.push_back(ctx->h_results);   // (hipHostFree waits for the device: see ensure)
    if (dev_env_on("MPX_ENSURE_FREE")) release_retired_blocks(ctx);
    ctx->h_results = nullptr;
    ctx->h_results_bytes = 0;
    void* p = nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    const hipError_t he = hipHostMalloc(&p, bytes + bytes / 4, hipHostMallocDefault);
    if (dev_env_on("MPX_ENSURE_TRACE"))
        fprintf(stderr, "mpx ensure: ctx %p hipHostMalloc(%zu) took %.0f us\n", (void*)ctx, bytes + bytes / 4,
                std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    if (he != hipSuccess) {
        (void)hipGetLastError();   // no pinned memory: the caller copies directly
        return nullptr;
    }
    ctx->h_results = p;
    ctx->h_results_bytes = bytes + bytes / 4;
    return p;
}

int d2h_results_sync(mpx_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes, hipStream_t st) {
    if (bytes > 8192) {
        if (void* h = pinned_results(ctx, bytes)) {
            MPX_HIP(ctx, hipMemcpyAsync(h, src_dev, bytes, hipMemcpyDeviceToHost, st));
            MPX_HIP(ctx, hipStreamSynchronize(st));
            std::memcpy(dst_host, h, bytes);
            return MPX_OK;
        }
    }
    MPX_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, st));
    MPX_HIP(ctx, hipStreamSynchronize(st));
    return MPX_OK;
}

void prof_mark_slow(mpx_ctx* ctx, hipStream_t st, const char* name) {
    if (ctx->prof_marks.size() >= (size_t)1 << 20) return;  // bounded: a forgotten mpx_profile_end must not eat the host
    hipEvent_t ev = nullptr;
    if (!ctx->prof_pool.empty()) {
        ev = ctx->prof_pool.back();
        ctx->prof_pool.pop_back();
    } else if (hipEventCreate(&ev) != hipSuccess) {
        return;
    }
    if (hipEventRecord(ev, st) != hipSuccess) {
        ctx->prof_pool.push_back(ev);
        return;
    }
    ctx->prof_marks.push_back({name, ev});
}

static int64_t num_frames_of(int64_t n, int frame, int hop) {
    if (n <= 0) return 0;
    if (hop == frame) return (n + frame - 1) / frame;
    if (n <= frame) return 1;
    return 1 + (n - frame + hop - 1) / hop;
}

// Frame descriptors for C clips packed back to back (each clip framed on its own).
static int build_descs(const int64_t* offsets, int num_clips, int frame, int hop,
                       std::vector<FrameDesc>& descs, std::vector<long long>& seg) {
    seg.assign(1, 0);
    for (int c = 0; c < num_clips; ++c) {
        const int64_t len = offsets[c + 1] - offsets[c];
        if (len < 0) return MPX_EINVAL;
        const int64_t nf = num_frames_of(len, frame, hop);
        for (int64_t f = 0; f < nf; ++f) {
            const int64_t s = f * hop;
            const int64_t left = len - s;
            descs.push_back({(long long)(offsets[c] + s), (int)(left >= frame ? frame : (left > 0 ? left : 0)), c});
        }
        seg.push_back((long long)descs.size());
    }
    return MPX_OK;
}

}  // namespace mpx

using namespace mpx;

namespace mpx {
// PCM_16 samples as a WAV file holds them -> the float32 the reference's loader hands the methods: x / 32768, exact in float32
// (include/mpx.h "PCM_16 input").  Eight samples per lane: one 16-byte load, two 16-byte stores.
__global__ __launch_bounds__(256) void pcm16_to_f32_kernel(const int16_t* __restrict__ pcm, long long n, float* __restrict__ out) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i + 8 <= n) {
        const int4 v = *reinterpret_cast<const int4*>(pcm + i);
        const int w[4] = {v.x, v.y, v.z, v.w};
        float f[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f[2 * k] = (float)(short)(w[k] & 0xffff) * (1.0f / 32768.0f);
            f[2 * k + 1] = (float)(short)(w[k] >> 16) * (1.0f / 32768.0f);
        }
        *reinterpret_cast<float4*>(out + i) = make_float4(f[0], f[1], f[2], f[3]);
        *reinterpret_cast<float4*>(out + i + 4) = make_float4(f[4], f[5], f[6], f[7]);
    } else {
        for (long long k = i; k < n; ++k) out[k] = (float)pcm[k] * (1.0f / 32768.0f);
    }
}
// the fit kernels' quotient and square root, element by element (mpx_test_lm_div_sqrt)
__global__ __launch_bounds__(256) void lm_div_sqrt_kernel(const double* __restrict__ a, const double* __restrict__ b, int n,
                                                          double* __restrict__ quot, double* __restrict__ root) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        quot[i] = mpx::lm::lm_div(a[i], b[i]);
        root[i] = mpx::lm::lm_sqrt(b[i]);
    }
}
}  // namespace mpx

extern "C" {

int mpx_abi_version(void) { return MPX_ABI_VERSION; }
int mpx_dev_knobs(void) { return mpx::DEV_KNOBS; }

int mpx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

mpx_ctx* mpx_create(int device, int flags) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        set_error(nullptr, MPX_EHIP, "no HIP device available (%s)", e == hipSuccess ? "count is 0" : hipGetErrorString(e));
        return nullptr;
    }
    if (device < 0 || device >= n) {
        set_error(nullptr, MPX_EINVAL, "device %d out of range [0,%d)", device, n);
        return nullptr;
    }
    if ((e = hipSetDevice(device)) != hipSuccess) {
        set_error(nullptr, MPX_EHIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
        return nullptr;
    }
    mpx_ctx* ctx = new (std::nothrow) mpx_ctx();
    if (!ctx) {
        set_error(nullptr, MPX_ENOMEM, "out of host memory");
        return nullptr;
    }
    ctx->device = device;
    ctx->flags = flags;
    if (getenv("MPX_DETERMINISTIC") && atoi(getenv("MPX_DETERMINISTIC"))) ctx->flags |= MPX_FLAG_DETERMINISTIC;  // read once
    // host batches >= 64 MiB go over PCIe in this many pieces (1 = one copy, no overlap); dev builds can change it, once
    ctx->copy_pieces = mpx::dev_env_on("MPX_NO_COPY_OVERLAP") ? 1 : mpx::dev_env_int("MPX_COPY_PIECES", 4);
    ctx->copy_pieces = ctx->copy_pieces < 1 ? 1 : (ctx->copy_pieces > 7 ? 7 : ctx->copy_pieces);
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) {
            ctx->num_cus = prop.multiProcessorCount;
            if (prop.totalGlobalMem / 8 > ctx->retired_cap) ctx->retired_cap = prop.totalGlobalMem / 8;   // (ensure, above)
        }
    }
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&ctx->ev0)) != hipSuccess || (e = hipEventCreate(&ctx->ev1)) != hipSuccess) {
        set_error(nullptr, MPX_EHIP, "stream/event creation failed: %s", hipGetErrorString(e));
        delete ctx;
        return nullptr;
    }
    return ctx;
}

void mpx_destroy(mpx_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    release_retired_blocks(ctx);
    for (void* p : ctx->owned) hipFree(p);
    for (DevBuf* b : {&ctx->d_pcm, &ctx->d_pcm_f32, &ctx->d_signal, &ctx->d_frames_out, &ctx->d_partials, &ctx->d_sum, &ctx->d_desc,
                      &ctx->d_offsets, &ctx->d_ws0, &ctx->d_ws1, &ctx->d_ws2, &ctx->d_ws3, &ctx->d_ws4, &ctx->d_counter, &ctx->d_queue})
        if (b->p) hipFree(b->p);
    for (auto& m : ctx->prof_marks) hipEventDestroy(m.ev);
    for (hipEvent_t e : ctx->prof_pool) hipEventDestroy(e);
    for (hipEvent_t e : ctx->copy_ev)
        if (e) hipEventDestroy(e);
    if (ctx->h_results) hipHostFree(ctx->h_results);
    if (ctx->h_tables) hipHostFree(ctx->h_tables);
    if (ctx->copy_stream) hipStreamDestroy(ctx->copy_stream);
    for (hipEvent_t e : {ctx->if0_ev_fe[0], ctx->if0_ev_fe[1], ctx->if0_ev_sp[0], ctx->if0_ev_sp[1]})
        if (e) hipEventDestroy(e);
    if (ctx->if0_sp_stream) {
        hipStreamSynchronize(ctx->if0_sp_stream);
        hipStreamDestroy(ctx->if0_sp_stream);
    }
    for (hipEvent_t e : ctx->side_ev)
        if (e) hipEventDestroy(e);
    if (ctx->side_stream) {
        hipStreamSynchronize(ctx->side_stream);
        hipStreamDestroy(ctx->side_stream);
    }
    if (ctx->ev0) hipEventDestroy(ctx->ev0);
    if (ctx->ev1) hipEventDestroy(ctx->ev1);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

int mpx_set_option(mpx_ctx* ctx, int option, int64_t value) {
    if (!ctx) return MPX_EINVAL;
    switch (option) {
        case MPX_OPT_IF0_WORKSPACE_BYTES:
            if (value < ((int64_t)64 << 20)) return set_error(ctx, MPX_EINVAL, "MPX_OPT_IF0_WORKSPACE_BYTES: %lld < 64 MiB", (long long)value);
            ctx->if0_ws_cap = (size_t)value;
            return MPX_OK;
        case MPX_OPT_HE_KERNEL:
            if (value != MPX_HE_KERNEL_AUTO && value != MPX_HE_KERNEL_WORKGROUP && value != MPX_HE_KERNEL_WAVE_PAIRS &&
                value != MPX_HE_KERNEL_WAVE_SERIAL)
                return set_error(ctx, MPX_EINVAL, "MPX_OPT_HE_KERNEL: unknown kernel %lld", (long long)value);
            ctx->he_kernel = (int)value;
            return MPX_OK;
    }
    return set_error(ctx, MPX_EINVAL, "unknown option %d", option);
}

int mpx_get_option(mpx_ctx* ctx, int option, int64_t* value) {
    if (!ctx || !value) return MPX_EINVAL;
    switch (option) {
        case MPX_OPT_IF0_WORKSPACE_BYTES: *value = (int64_t)ctx->if0_ws_cap; return MPX_OK;
        case MPX_OPT_HE_KERNEL: *value = ctx->he_kernel; return MPX_OK;
    }
    return set_error(ctx, MPX_EINVAL, "unknown option %d", option);
}

void* mpx_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

void mpx_host_free(void* p) {
    if (p) hipHostFree(p);
}

const char* mpx_last_error(const mpx_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

unsigned mpx_launch_count(const mpx_ctx* ctx) { return ctx ? ctx->launches.load(std::memory_order_relaxed) : 0u; }

int mpx_synchronize(mpx_ctx* ctx) {
    if (!ctx) return MPX_EINVAL;
    MPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MPX_OK;
}

void* mpx_stream(mpx_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int64_t mpx_num_frames(int64_t n, int frame, int hop) {
    if (frame <= 0 || hop <= 0 || hop > frame) return -1;
    return num_frames_of(n, frame, hop);
}

int mpx_timer_begin(mpx_ctx* ctx, void* stream) {
    if (!ctx) return MPX_EINVAL;
    MPX_HIP(ctx, hipEventRecord(ctx->ev0, stream ? (hipStream_t)stream : ctx->stream));
    return MPX_OK;
}

int mpx_timer_end(mpx_ctx* ctx, void* stream, float* ms) {
    if (!ctx || !ms) return MPX_EINVAL;
    MPX_HIP(ctx, hipEventRecord(ctx->ev1, stream ? (hipStream_t)stream : ctx->stream));
    MPX_HIP(ctx, hipEventSynchronize(ctx->ev1));
    MPX_HIP(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return MPX_OK;
}

int mpx_profile_begin(mpx_ctx* ctx) {
    if (!ctx) return MPX_EINVAL;
    for (auto& m : ctx->prof_marks) ctx->prof_pool.push_back(m.ev);
    ctx->prof_marks.clear();
    ctx->fit_stats[0] = ctx->fit_stats[1] = ctx->fit_stats[2] = 0;
    ctx->prof_on = true;
    return MPX_OK;
}

int mpx_esacf_fit_stats(mpx_ctx* ctx, int64_t* stats3) {
    if (!ctx || !stats3) return MPX_EINVAL;
    for (int i = 0; i < 3; ++i) stats3[i] = ctx->fit_stats[i];
    return MPX_OK;
}

int mpx_profile_end(mpx_ctx* ctx, char* report, int cap) {
    if (!ctx || !report || cap < 1) return MPX_EINVAL;
    ctx->prof_on = false;
    report[0] = 0;
    std::map<std::string, std::pair<long long, double>> acc;
    std::vector<std::string> order;
    for (size_t i = 0; i + 1 < ctx->prof_marks.size(); ++i) {
        const auto& a = ctx->prof_marks[i];
        if (!a.name) continue;
        MPX_HIP(ctx, hipEventSynchronize(ctx->prof_marks[i + 1].ev));
        float ms = 0.f;
        MPX_HIP(ctx, hipEventElapsedTime(&ms, a.ev, ctx->prof_marks[i + 1].ev));
        auto it = acc.find(a.name);
        if (it == acc.end()) {
            order.push_back(a.name);
            it = acc.emplace(a.name, std::make_pair(0LL, 0.0)).first;
        }
        it->second.first += 1;
        it->second.second += ms;
    }
    std::string out;
    for (const auto& name : order) {
        char line[256];
        snprintf(line, sizeof line, "%s %lld %.6f\n", name.c_str(), acc[name].first, acc[name].second);
        out += line;
    }
    for (auto& m : ctx->prof_marks) ctx->prof_pool.push_back(m.ev);
    ctx->prof_marks.clear();
    if ((int)out.size() + 1 > cap) return set_error(ctx, MPX_EINVAL, "mpx_profile_end: report needs %zu bytes", out.size() + 1);
    memcpy(report, out.c_str(), out.size() + 1);
    return MPX_OK;
}

// ------------------------------------------------------------------ method 2
static int check_common(mpx_ctx* ctx, const void* sig, int64_t n, int frame, int hop) {
    if (!ctx) return MPX_EINVAL;
    ctx->err.clear();
    if (n < 0 || (n > 0 && !sig)) return set_error(ctx, MPX_EINVAL, "signal pointer/length invalid");
    if (frame <= 0 || hop <= 0 || hop > frame)
        return set_error(ctx, MPX_EINVAL, "need 0 < hop <= frame (got frame=%d hop=%d)", frame, hop);
    MPX_HIP(ctx, hipSetDevice(ctx->device));
    return MPX_OK;
}

// A method runner fills d_frames ([F,12], may be NULL when the method can do without) and, when
// it owns a fused cross-frame reduction, d_sum; *did_sum tells the caller whether it did.
typedef int (*run_fn)(mpx_ctx*, const float*, int64_t, const FrameDesc*, int64_t, int, const void*, int, int,
                      double*, double*, bool*, hipStream_t);

static int run_he(mpx_ctx* c, const float* s, int64_t n, const FrameDesc* d, int64_t nf, int fs, const void* p,
                  int frame, int hop, double* out, double* sum, bool* did_sum, hipStream_t st) {
    // clips (desc mode) are reduced per segment by the caller; a single signal is reduced in the kernel
    double* fused = d ? nullptr : sum;
    *did_sum = fused != nullptr;
    return he_run(c, s, n, d, nf, fs, (const mpx_he_params*)p, frame, hop, out, fused, st);
}
static int run_esacf(mpx_ctx* c, const float* s, int64_t n, const FrameDesc* d, int64_t nf, int fs, const void* p,
                     int frame, int hop, double* out, double* sum, bool* did_sum, hipStream_t st) {
    *did_sum = false;
    if (!out) {  // ESACF always goes through per-frame rows
        int rc = ensure(c, c->d_frames_out, (size_t)(nf ? nf : 1) * 12 * sizeof(double));
        if (rc) return rc;
        out = (double*)c->d_frames_out.p;
    }
    int rc = esacf_run(c, s, n, d, nf, fs, (const mpx_esacf_params*)p, frame, hop, out, -1, nullptr, st);
    if (rc) return rc;
    if (sum && !d && nf) {
        *did_sum = true;
        return segment_sum(c, out, nullptr, 1, nf, sum, st);
    }
    return MPX_OK;
}

// device-resident single signal
static int method_dev(mpx_ctx* ctx, run_fn run, const float* d_signal, int64_t n, int fs, const void* params,
                      int frame, int hop, double* d_chroma_frames, double* d_chroma_sum, void* stream) {
    int rc = check_common(ctx, d_signal, n, frame, hop);
    if (rc) return rc;
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    const int64_t nf = num_frames_of(n, frame, hop);
    bool did_sum = false;
    if (nf && (rc = run(ctx, d_signal, n, nullptr, nf, fs, params, frame, hop, d_chroma_frames, d_chroma_sum, &did_sum, st)))
        return rc;
    if (d_chroma_sum && !did_sum) {
        if (nf == 0)
            MPX_HIP(ctx, hipMemsetAsync(d_chroma_sum, 0, 12 * sizeof(double), st));
        else
            return set_error(ctx, MPX_EHIP, "internal: method did not reduce its frames");
    }
    return MPX_OK;
}

// host single signal
// The copy stream and its eight events exist together or not at all: a partial failure is undone, and the call (and the
// next one: it tries again) goes in one piece.
static bool copy_stream_ready(mpx_ctx* ctx) {
    if (ctx->copy_ready) return true;
    bool ok = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) == hipSuccess;
    int made = 0;
    for (; ok && made < 8; ++made) ok = hipEventCreateWithFlags(&ctx->copy_ev[made], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        for (int k = 0; k < 8; ++k) {
            if (ctx->copy_ev[k]) (void)hipEventDestroy(ctx->copy_ev[k]);
            ctx->copy_ev[k] = nullptr;
        }
        if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
        ctx->copy_stream = nullptr;
        return false;
    }
    ctx->copy_ready = true;
    return true;
}

static void pcm16_convert(const int16_t* d_pcm, long long n, float* d_out, hipStream_t st) {
    const long long groups = (n + 7) / 8;
    if (groups > 0)
        hipLaunchKernelGGL(pcm16_to_f32_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st, d_pcm, n, d_out);
}

// int16 samples (host or device memory) -> float32 in `d_out` (device), on `st`: the copy moves 2 bytes per sample
static int pcm16_stage(mpx_ctx* ctx, const int16_t* pcm, int64_t n, float* d_out, hipStream_t st) {
    if (!n) return MPX_OK;
    int rc = ensure(ctx, ctx->d_pcm, (size_t)n * sizeof(int16_t) + 16);
    if (rc) return rc;
    if ((rc = stage_h2d(ctx, ctx->d_pcm.p, pcm, (size_t)n * sizeof(int16_t), st))) return rc;
    pcm16_convert((const int16_t*)ctx->d_pcm.p, (long long)n, d_out, st);
    MPX_HIP(ctx, hipGetLastError());
    return MPX_OK;
}

static int method_host(mpx_ctx* ctx, run_fn run, const void* signal_any, bool pcm16, int64_t n, int fs, const void* params,
                       int frame, int hop, double* chroma_frames, double* chroma_sum) {
    const float* signal = (const float*)signal_any;
    int rc = check_common(ctx, signal, n, frame, hop);
    if (rc) return rc;
    if (!chroma_sum) return set_error(ctx, MPX_EINVAL, "chroma_sum must not be NULL");
    const int64_t nf = num_frames_of(n, frame, hop);
    const bool on_device = !pcm16 && n && samples_on_device(ctx, signal);   // float32 samples already in HBM: read in place (include/mpx.h)
    if (!on_device && (rc = ensure(ctx, ctx->d_signal, (size_t)(n ? n : 1) * sizeof(float)))) return rc;
    if ((rc = ensure(ctx, ctx->d_frames_out, (size_t)(nf ? nf : 1) * 12 * sizeof(double)))) return rc;
    if ((rc = ensure(ctx, ctx->d_sum, 12 * sizeof(double)))) return rc;
    // (Round 6 measured the copy of a long host signal in four pieces with the frames that had arrived computed under the next
    // piece's copy -- rows and sum bit-equal to this path: SLOWER, 0.69 -> 0.73 ms for the 33.5 MB headline signal and 0.39 ->
    // 0.44 ms as PCM_16: the 0.04 ms of kernels it can hide are less than four copies, events and launches cost;
    // profiles/r6/h2d_probe_copy_in_pieces_rejected.json.  Removed.)
    if (pcm16) {
        if ((rc = pcm16_stage(ctx, (const int16_t*)signal_any, n, (float*)ctx->d_signal.p, ctx->stream))) return rc;
    } else if (n && !on_device && (rc = stage_h2d(ctx, ctx->d_signal.p, signal, (size_t)n * sizeof(float), ctx->stream))) {
        return rc;
    }
    rc = method_dev(ctx, run, on_device ? signal : (const float*)ctx->d_signal.p, n, fs, params, frame, hop,
                    chroma_frames ? (double*)ctx->d_frames_out.p : nullptr, (double*)ctx->d_sum.p, ctx->stream);
    if (rc) return rc;
    MPX_HIP(ctx, hipMemcpyAsync(chroma_sum, ctx->d_sum.p, 12 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (chroma_frames && nf)
        MPX_HIP(ctx, hipMemcpyAsync(chroma_frames, ctx->d_frames_out.p, (size_t)nf * 12 * sizeof(double),
                                    hipMemcpyDeviceToHost, ctx->stream));
    MPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MPX_OK;
}

// host batch of clips
static int method_batch(mpx_ctx* ctx, run_fn run, const float* signals, const int64_t* offsets, int num_clips,
                        int fs, const void* params, int frame, int hop, double* chroma_sums) {
    if (!ctx) return MPX_EINVAL;
    if (num_clips < 0 || !offsets || (num_clips > 0 && !chroma_sums))
        return set_error(ctx, MPX_EINVAL, "bad batch arguments");
    if (num_clips == 0) return MPX_OK;
    dev_tick(ctx, "method_batch: enter");
    const int64_t total = offsets[num_clips];
    int rc = check_common(ctx, signals, total, frame, hop);
    if (rc) return rc;
    if (offsets[0] != 0) return set_error(ctx, MPX_EINVAL, "offsets[0] must be 0");
    // the frame descriptors and the segment table of this layout may still be on the device from the previous call
    std::vector<int64_t> layout(offsets, offsets + num_clips + 1);
    layout.push_back(frame);
    layout.push_back(hop);
    const bool cached = layout == ctx->batch_layout && ctx->d_desc.p && ctx->d_offsets.p;
    std::vector<FrameDesc> descs;
    std::vector<long long> seg;
    if (!cached && build_descs(offsets, num_clips, frame, hop, descs, seg))
        return set_error(ctx, MPX_EINVAL, "offsets must be non-decreasing");
    const int64_t nf = cached ? ctx->batch_layout_frames : (int64_t)descs.size();
    dev_tick(ctx, "method_batch: descriptors");
    ctx->batch_layout.clear();
    // where the samples live (include/mpx.h): clips already in HBM are read IN PLACE, no copy into the context's buffer
    const bool on_device = total && samples_on_device(ctx, signals);
    if (!on_device && (rc = ensure(ctx, ctx->d_signal, (size_t)(total ? total : 1) * sizeof(float)))) return rc;
    const float* d_in = on_device ? signals : (const float*)ctx->d_signal.p;
    if ((rc = ensure(ctx, ctx->d_frames_out, (size_t)(nf ? nf : 1) * 12 * sizeof(double)))) return rc;
    if ((rc = ensure(ctx, ctx->d_desc, (size_t)(nf ? nf : 1) * sizeof(FrameDesc)))) return rc;
    if ((rc = ensure(ctx, ctx->d_offsets, (size_t)(num_clips + 1) * sizeof(long long)))) return rc;
    if ((rc = ensure(ctx, ctx->d_sum, (size_t)num_clips * 12 * sizeof(double)))) return rc;
    hipStream_t st = ctx->stream;
    dev_tick(ctx, "method_batch: workspaces");
    if (!cached) {   // (the buffers only grow: a layout that was cached fits them as they are)
        if (nf) MPX_HIP(ctx, hipMemcpyAsync(ctx->d_desc.p, descs.data(), (size_t)nf * sizeof(FrameDesc), hipMemcpyHostToDevice, st));
        MPX_HIP(ctx, hipMemcpyAsync(ctx->d_offsets.p, seg.data(), seg.size() * sizeof(long long), hipMemcpyHostToDevice, st));
    }
    dev_tick(ctx, "method_batch: tables queued");
    bool did_sum = false;
    // A large batch in HOST memory goes over PCIe in pieces on a second stream, the kernels of piece k running next to
    // the copy of piece k+1 (ESACF, 4096 clips: the 1.4 GB copy and the 27 ms of kernels take about as long as each
    // other).  Device-resident batches and small ones: one copy, one pass.
    int pieces = 1;
    if (!on_device && (size_t)total * sizeof(float) >= (size_t(64) << 20) && num_clips >= 8) pieces = ctx->copy_pieces;
    if (pieces > 1 && cached && build_descs(offsets, num_clips, frame, hop, descs, seg))   // the pieces' frame ranges come from the host's table
        return set_error(ctx, MPX_EINVAL, "offsets must be non-decreasing");
    if (pieces > 1 && !copy_stream_ready(ctx)) pieces = 1;
    if (pieces == 1) {
        if (total && !on_device && (rc = stage_h2d(ctx, ctx->d_signal.p, signals, (size_t)total * sizeof(float), st))) return rc;
        if (nf && (rc = run(ctx, d_in, total, (const FrameDesc*)ctx->d_desc.p, nf, fs, params,
                            frame, hop, (double*)ctx->d_frames_out.p, nullptr, &did_sum, st)))
            return rc;
    } else {
        // the staging buffer may still be read by work queued earlier on the compute stream
        MPX_HIP(ctx, hipEventRecord(ctx->copy_ev[7], st));
        MPX_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->copy_ev[7], 0));
        int c0 = 0;
        int bounds[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        bounds[pieces] = num_clips;
        for (int k = 1; k < pieces; ++k) {   // clip boundaries nearest to k/pieces of the samples
            const int64_t want = total / pieces * k;
            while (c0 < num_clips && offsets[c0] < want) ++c0;
            bounds[k] = c0;
        }
        for (int k = 0; k < pieces; ++k) {
            // copy of piece k, then the (asynchronous) kernels of piece k: a copy from pageable memory keeps the CALLING
            // THREAD busy until it is done, so it is the next piece's copy that runs next to these kernels
            const int64_t s0 = offsets[bounds[k]], s1 = offsets[bounds[k + 1]];
            if (s1 > s0)
                MPX_HIP(ctx, hipMemcpyAsync((float*)ctx->d_signal.p + s0, signals + s0, (size_t)(s1 - s0) * sizeof(float),
                                            hipMemcpyDefault, ctx->copy_stream));
            MPX_HIP(ctx, hipEventRecord(ctx->copy_ev[k], ctx->copy_stream));
            const long long f0 = seg[bounds[k]], f1 = seg[bounds[k + 1]];
            MPX_HIP(ctx, hipStreamWaitEvent(st, ctx->copy_ev[k], 0));
            if (f1 > f0 && (rc = run(ctx, (const float*)ctx->d_signal.p, total, (const FrameDesc*)ctx->d_desc.p + f0, f1 - f0,
                                     fs, params, frame, hop, (double*)ctx->d_frames_out.p + f0 * 12, nullptr, &did_sum, st)))
                return rc;
        }
    }
    dev_tick(ctx, "method_batch: kernels queued");
    // The per-clip sums are written by the kernel straight into the context's pinned (device-mapped) staging when there is one: a
    // copy of its own costs a launch and its latency (round 6: 148 -> 140 us per 1366-clip Harmonic-Energy call, same-box A/B with
    // MPX_RESULTS_BY_COPY=1 in the development library, profiles/r6/batch_results_in_place_ab.txt).
    const size_t rbytes = (size_t)num_clips * 12 * sizeof(double);
    void* hres = rbytes > 8192 && !dev_env_on("MPX_RESULTS_BY_COPY") ? pinned_results(ctx, rbytes) : nullptr;
    if ((rc = segment_sum(ctx, (const double*)ctx->d_frames_out.p, (const long long*)ctx->d_offsets.p, num_clips, nf,
                          hres ? (double*)hres : (double*)ctx->d_sum.p, st)))
        return rc;
    if (hres) {
        MPX_HIP(ctx, hipStreamSynchronize(st));
        std::memcpy(chroma_sums, hres, rbytes);
    } else if ((rc = d2h_results_sync(ctx, chroma_sums, ctx->d_sum.p, rbytes, st))) {
        return rc;
    }
    ctx->batch_layout.swap(layout);   // d_desc / d_offsets hold this layout now (a failed call leaves the cache empty)
    ctx->batch_layout_frames = nf;
    return MPX_OK;
}

int mpx_harmonic_energy(mpx_ctx* ctx, const float* signal, int64_t n, int fs, const mpx_he_params* params,
                        int frame, int hop, double* chroma_frames, double* chroma_sum) {
    return method_host(ctx, run_he, signal, false, n, fs, params, frame, hop, chroma_frames, chroma_sum);
}

int mpx_harmonic_energy_batch(mpx_ctx* ctx, const float* signals, const int64_t* offsets, int num_clips, int fs,
                              const mpx_he_params* params, int frame, int hop, double* chroma_sums) {
    return method_batch(ctx, run_he, signals, offsets, num_clips, fs, params, frame, hop, chroma_sums);
}

int mpx_harmonic_energy_dev(mpx_ctx* ctx, const float* d_signal, int64_t n, int fs, const mpx_he_params* params,
                            int frame, int hop, double* d_chroma_frames, double* d_chroma_sum, void* stream) {
    return method_dev(ctx, run_he, d_signal, n, fs, params, frame, hop, d_chroma_frames, d_chroma_sum, stream);
}

int mpx_harmonic_energy_argmax(mpx_ctx* ctx, const float* signal, int64_t n, int fs, const mpx_he_params* params, int frame,
                               int hop, int32_t* best_ind, int32_t* bounds) {
    int rc = check_common(ctx, signal, n, frame, hop);
    if (rc) return rc;
    const int64_t nf = num_frames_of(n, frame, hop);
    if (nf && !best_ind) return set_error(ctx, MPX_EINVAL, "best_ind must not be NULL");
    const mpx_he_params p = params ? *params : mpx_he_params{2, 2, 2};
    if (nf == 0) return he_argmax_run(ctx, nullptr, 0, 0, fs, &p, frame, hop, nullptr, bounds, ctx->stream);
    if (p.num_harmonic < 1 || p.num_octave < 1 || p.num_harmonic * p.num_octave > 64)
        return set_error(ctx, MPX_EINVAL, "bad harmonic-energy params (%d,%d,%d)", p.num_harmonic, p.num_octave, p.num_bins);
    const size_t nwin = (size_t)12 * p.num_octave * p.num_harmonic;
    if ((rc = ensure(ctx, ctx->d_signal, (size_t)n * sizeof(float)))) return rc;
    if ((rc = ensure(ctx, ctx->d_ws0, (size_t)nf * nwin * sizeof(int32_t)))) return rc;
    if ((rc = stage_h2d(ctx, ctx->d_signal.p, signal, (size_t)n * sizeof(float), ctx->stream))) return rc;
    if ((rc = he_argmax_run(ctx, (const float*)ctx->d_signal.p, n, nf, fs, &p, frame, hop, (int*)ctx->d_ws0.p, bounds, ctx->stream))) return rc;
    MPX_HIP(ctx, hipMemcpyAsync(best_ind, ctx->d_ws0.p, (size_t)nf * nwin * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    MPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MPX_OK;
}

// ------------------------------------------------------------------ method 1
int mpx_esacf(mpx_ctx* ctx, const float* signal, int64_t n, int fs, const mpx_esacf_params* params, int frame,
              int hop, double* chroma_frames, double* chroma_sum) {
    return method_host(ctx, run_esacf, signal, false, n, fs, params, frame, hop, chroma_frames, chroma_sum);
}

int mpx_esacf_batch(mpx_ctx* ctx, const float* signals, const int64_t* offsets, int num_clips, int fs,
                    const mpx_esacf_params* params, int frame, int hop, double* chroma_sums) {
    return method_batch(ctx, run_esacf, signals, offsets, num_clips, fs, params, frame, hop, chroma_sums);
}

int mpx_esacf_dev(mpx_ctx* ctx, const float* d_signal, int64_t n, int fs, const mpx_esacf_params* params,
                  int frame, int hop, double* d_chroma_frames, double* d_chroma_sum, void* stream) {
    return method_dev(ctx, run_esacf, d_signal, n, fs, params, frame, hop, d_chroma_frames, d_chroma_sum, stream);
}

// ------------------------------------------------------------------ PCM_16 input (ABI 6): include/mpx.h
int mpx_harmonic_energy_pcm16(mpx_ctx* ctx, const int16_t* pcm, int64_t n, int fs, const mpx_he_params* params, int frame,
                              int hop, double* chroma_frames, double* chroma_sum) {
    return method_host(ctx, run_he, pcm, true, n, fs, params, frame, hop, chroma_frames, chroma_sum);
}

int mpx_esacf_pcm16(mpx_ctx* ctx, const int16_t* pcm, int64_t n, int fs, const mpx_esacf_params* params, int frame, int hop,
                    double* chroma_frames, double* chroma_sum) {
    return method_host(ctx, run_esacf, pcm, true, n, fs, params, frame, hop, chroma_frames, chroma_sum);
}

// methods 3 and 4 take their samples wherever they live: converted into a buffer of the context's, then handed over as
// device memory (their runners read device-resident clips in place or copy them inside HBM)
static int pcm16_to_ctx(mpx_ctx* ctx, const int16_t* pcm, int64_t n, const float** d_out) {
    if (!ctx) return MPX_EINVAL;
    ctx->err.clear();
    if (n < 0 || (n > 0 && !pcm)) return set_error(ctx, MPX_EINVAL, "signal pointer/length invalid");
    MPX_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure(ctx, ctx->d_pcm_f32, (size_t)(n ? n : 1) * sizeof(float));
    if (rc) return rc;
    if ((rc = pcm16_stage(ctx, pcm, n, (float*)ctx->d_pcm_f32.p, ctx->stream))) return rc;
    MPX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the runners' own copies are ordered on their streams, not behind this one
    *d_out = (const float*)ctx->d_pcm_f32.p;
    return MPX_OK;
}

int mpx_prime_multif0_pcm16(mpx_ctx* ctx, const int16_t* pcm, int64_t n, int fs, const mpx_prime_params* params,
                            double* chroma_sum) {
    const float* d = nullptr;
    int rc = pcm16_to_ctx(ctx, pcm, n, &d);
    if (rc) return rc;
    return mpx_prime_multif0(ctx, d, n, fs, params, chroma_sum);
}

int mpx_iterative_f0_pcm16(mpx_ctx* ctx, const int16_t* pcm, int64_t n, int fs, const mpx_if0_params* params,
                           double* chroma_frames, double* chroma_sum) {
    const float* d = nullptr;
    int rc = pcm16_to_ctx(ctx, pcm, n, &d);
    if (rc) return rc;
    return mpx_iterative_f0(ctx, d, n, fs, params, chroma_frames, chroma_sum);
}

// ------------------------------------------------------------------ method 3
static int if0_common(mpx_ctx* ctx, const float* signals, const int64_t* offsets, int num_clips) {
    if (!ctx) return MPX_EINVAL;
    ctx->err.clear();
    if (num_clips < 0 || !offsets) return set_error(ctx, MPX_EINVAL, "bad batch arguments");
    if (num_clips && offsets[0] != 0) return set_error(ctx, MPX_EINVAL, "offsets[0] must be 0");
    if (num_clips && offsets[num_clips] > 0 && !signals) return set_error(ctx, MPX_EINVAL, "signal pointer/length invalid");
    MPX_HIP(ctx, hipSetDevice(ctx->device));
    return MPX_OK;
}

int mpx_iterative_f0_batch(mpx_ctx* ctx, const float* signals, const int64_t* offsets, int num_clips, int fs,
                           const mpx_if0_params* params, double* chroma_sums) {
    int rc = if0_common(ctx, signals, offsets, num_clips);
    if (rc) return rc;
    if (num_clips == 0) return MPX_OK;
    if (!chroma_sums) return set_error(ctx, MPX_EINVAL, "chroma_sums must not be NULL");
    return if0_run_host(ctx, signals, offsets, num_clips, fs, params, nullptr, chroma_sums, nullptr);
}

int mpx_iterative_f0(mpx_ctx* ctx, const float* signal, int64_t n, int fs, const mpx_if0_params* params,
                     double* chroma_frames, double* chroma_sum) {
    if (!ctx) return MPX_EINVAL;
    if (n < 0 || !chroma_sum) return set_error(ctx, MPX_EINVAL, "bad arguments");
    const int64_t offsets[2] = {0, n};
    int rc = if0_common(ctx, signal, offsets, 1);
    if (rc) return rc;
    return if0_run_host(ctx, signal, offsets, 1, fs, params, chroma_frames, chroma_sum, nullptr);
}

int mpx_iterative_f0_periodicity(mpx_ctx* ctx, const double* spectra, int64_t num_frames, int bins, int fs,
                                 const mpx_if0_params* params, double* chroma_frames) {
    if (!ctx) return MPX_EINVAL;
    ctx->err.clear();
    MPX_HIP(ctx, hipSetDevice(ctx->device));
    return if0_periodicity_host(ctx, spectra, num_frames, bins, fs, params, chroma_frames);
}

int mpx_iterative_f0_periodicity_voices(mpx_ctx* ctx, const double* spectra, int64_t num_frames, int bins, int fs,
                                        const mpx_if0_params* params, double* chroma_frames, double* saliences, double* periods) {
    if (!ctx) return MPX_EINVAL;
    ctx->err.clear();
    MPX_HIP(ctx, hipSetDevice(ctx->device));
    return if0_periodicity_host(ctx, spectra, num_frames, bins, fs, params, chroma_frames, saliences, periods);
}

int mpx_iterative_f0_dev(mpx_ctx* ctx, const float* d_signal, int64_t n, int fs, const mpx_if0_params* params,
                         double* d_chroma_frames, double* d_chroma_sum, void* stream) {
    if (!ctx) return MPX_EINVAL;
    if (n < 0 || (!d_chroma_frames && !d_chroma_sum)) return set_error(ctx, MPX_EINVAL, "bad arguments");
    const int64_t offsets[2] = {0, n};
    int rc = if0_common(ctx, d_signal, offsets, 1);
    if (rc) return rc;
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    if (n == 0) {
        if (d_chroma_sum) MPX_HIP(ctx, hipMemsetAsync(d_chroma_sum, 0, 12 * sizeof(double), st));
        return MPX_OK;
    }
    return if0_run_host(ctx, d_signal, offsets, 1, fs, params, d_chroma_frames, d_chroma_sum, nullptr, true, st);
}

int mpx_iterative_f0_spectra(mpx_ctx* ctx, const float* signal, int64_t n, int fs, const mpx_if0_params* params,
                             double* ut) {
    if (!ctx) return MPX_EINVAL;
    if (n < 0 || !ut) return set_error(ctx, MPX_EINVAL, "bad arguments");
    const int64_t offsets[2] = {0, n};
    int rc = if0_common(ctx, signal, offsets, 1);
    if (rc) return rc;
    return if0_run_host(ctx, signal, offsets, 1, fs, params, nullptr, nullptr, ut);
}

int mpx_iterative_f0_warmup(mpx_ctx* ctx, int fs, const mpx_if0_params* params, int64_t* samples, double* pole_radius) {
    if (!ctx || !samples) return MPX_EINVAL;
    long long w = 0;
    int rc = if0_warmup_samples(ctx, fs, params, &w, pole_radius);
    if (rc) return rc;
    *samples = (int64_t)w;
    return MPX_OK;
}

// ------------------------------------------------------------------ method 4
int mpx_prime_multif0_batch(mpx_ctx* ctx, const float* signals, const int64_t* offsets, int num_clips, int fs,
                            const mpx_prime_params* params, double* chroma_sums) {
    if (!ctx) return MPX_EINVAL;
    ctx->err.clear();
    if (num_clips < 0 || !offsets || (num_clips > 0 && !chroma_sums)) return set_error(ctx, MPX_EINVAL, "bad batch arguments");
    if (num_clips == 0) return MPX_OK;
    if (offsets[0] != 0) return set_error(ctx, MPX_EINVAL, "offsets[0] must be 0");
    if (offsets[num_clips] > 0 && !signals) return set_error(ctx, MPX_EINVAL, "signal pointer/length invalid");
    MPX_HIP(ctx, hipSetDevice(ctx->device));
    return prime_run_host(ctx, signals, offsets, num_clips, fs, params, chroma_sums);
}

int mpx_prime_multif0_dev(mpx_ctx* ctx, const float* d_signal, int64_t n, int fs, const mpx_prime_params* params,
                          double* d_chroma_sum, void* stream) {
    if (!ctx) return MPX_EINVAL;
    ctx->err.clear();
    if (n < 0 || !d_chroma_sum || (n > 0 && !d_signal)) return set_error(ctx, MPX_EINVAL, "bad arguments");
    MPX_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t offsets[2] = {0, n};
    return prime_run_host(ctx, d_signal, offsets, 1, fs, params, d_chroma_sum, true, stream ? (hipStream_t)stream : ctx->stream);
}

int mpx_prime_multif0(mpx_ctx* ctx, const float* signal, int64_t n, int fs, const mpx_prime_params* params,
                      double* chroma_sum) {
    if (!ctx) return MPX_EINVAL;
    if (n < 0 || !chroma_sum) return set_error(ctx, MPX_EINVAL, "bad arguments");
    const int64_t offsets[2] = {0, n};
    return mpx_prime_multif0_batch(ctx, signal, offsets, 1, fs, params, chroma_sum);
}

int mpx_set_remez_taps(mpx_ctx* ctx, int fs, const double* taps13) {
    if (!ctx || !taps13 || fs <= 0) return MPX_EINVAL;
    ctx->remez[fs] = std::vector<double>(taps13, taps13 + 13);
    ctx->host_blobs.erase("bs_runin_" + std::to_string(fs));   // the band splitter's run-in (band_cut, mpx_esacf.hip) was simulated with the taps before
    return MPX_OK;
}

// host-callable copy of the device peak fit, for CPU-side unit tests of the restatement
int mpx_test_gaussian_fit(const double* xs, const double* ys, int m, double* center) {
    if (!xs || !ys || !center || m < 3 || m > mpx::lm::MAXM) return MPX_EINVAL;
    mpx::lm::Problem pr;
    pr.m = m;
    for (int i = 0; i < m; ++i) {
        pr.xs[i] = xs[i];
        pr.ys[i] = ys[i];
    }
    return mpx::lm::gaussian_fit(pr, center);
}

// host-callable copy of the SACF kernels' |X|^0.67
int mpx_test_pow067(const double* x, int n, double* out) {
    if (!x || !out || n < 0) return MPX_EINVAL;
    double tab[mpx::p067::TAB_DOUBLES];
    mpx::p067::build_tables(tab);
    for (int i = 0; i < n; ++i) out[i] = mpx::p067::pow067(x[i], tab);
    return MPX_OK;
}

// the fit kernels' quotient and square root on the device, element by element (tests/test_gpu_lm_div_sqrt.py)
int mpx_test_lm_div_sqrt(mpx_ctx* ctx, const double* a, const double* b, int n, double* quot, double* root) {
    if (!ctx) return MPX_EINVAL;
    if (!a || !b || !quot || !root || n < 0) return set_error(ctx, MPX_EINVAL, "NULL pointer or negative count");
    if (n == 0) return MPX_OK;
    const size_t bytes = (size_t)n * sizeof(double);
    double* d = nullptr;
    if (hipMalloc(&d, 4 * bytes) != hipSuccess) return set_error(ctx, MPX_ENOMEM, "hipMalloc(%zu) failed", 4 * bytes);
    int rc = MPX_OK;
    if (hipMemcpyAsync(d, a, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(d + n, b, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
        rc = set_error(ctx, MPX_EHIP, "copy to the device failed");
    } else {
        mpx::lm_div_sqrt_kernel<<<(n + 255) / 256, 256, 0, ctx->stream>>>(d, d + n, n, d + 2 * (size_t)n, d + 3 * (size_t)n);
        if (hipMemcpyAsync(quot, d + 2 * (size_t)n, bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(root, d + 3 * (size_t)n, bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess)
            rc = set_error(ctx, MPX_EHIP, "kernel or copy back failed: %s", hipGetErrorString(hipGetLastError()));
    }
    (void)hipFree(d);
    return rc;
}

int mpx_esacf_stage(mpx_ctx* ctx, int stage, const float* signal, int64_t n, int fs,
                    const mpx_esacf_params* params, int frame, int hop, double* out) {
    int rc = check_common(ctx, signal, n, frame, hop);
    if (rc) return rc;
    if (stage < MPX_STAGE_WFIR || stage > MPX_STAGE_ESACF || !out)
        return set_error(ctx, MPX_EINVAL, "bad stage id %d or NULL out", stage);
    const int64_t nf = num_frames_of(n, frame, hop);
    if (nf == 0) return MPX_OK;
    const size_t len = stage <= MPX_STAGE_XHI ? (size_t)frame : (size_t)((frame - 1) / 2);
    if ((rc = ensure(ctx, ctx->d_signal, (size_t)n * sizeof(float)))) return rc;
    if ((rc = ensure(ctx, ctx->d_frames_out, (size_t)nf * 12 * sizeof(double)))) return rc;
    if ((rc = ensure(ctx, ctx->d_ws2, (size_t)nf * len * sizeof(double) + 16))) return rc;
    if ((rc = stage_h2d(ctx, ctx->d_signal.p, signal, (size_t)n * sizeof(float), ctx->stream))) return rc;
    if ((rc = esacf_run(ctx, (const float*)ctx->d_signal.p, n, nullptr, nf, fs, params, frame, hop,
                        (double*)ctx->d_frames_out.p, stage, (double*)ctx->d_ws2.p, ctx->stream)))
        return rc;
    MPX_HIP(ctx, hipMemcpyAsync(out, ctx->d_ws2.p, (size_t)nf * len * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MPX_OK;
}

}  // extern "C"
