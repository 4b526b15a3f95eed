/* mpx.h -- C ABI of the MI355X-native multipitch / chromagram engine.
 *
 * The reference (sevagh/chord-detection) is pure Python and has no FFI of its
 * own; its operator interface for this path is the `Multipitch` ABC
 * (chord_detection/multipitch.py:9-44) and the `Chromagram` value type
 * (chord_detection/chromagram.py:11-58).  This header is the boundary a
 * maintainer binds with ctypes to replace the NumPy/SciPy
 * bodies of
 *     MultipitchHarmonicEnergy.compute_pitches   harmonic_energy.py:30-73
 *     MultipitchESACF.compute_pitches            esacf.py:41-91
 *     dsp.frame_cutter / wfir / lowpass_filter   dsp/frame.py:5-14, dsp/wfir.py:25-43,
 *                                                dsp/lowpass.py:6-8, esacf.py:132-134
 * Everything here is plain C: pointers, sizes, PODs.  No torch types.
 *
 * Conventions
 *   - every call returns MPX_OK (0) or a negative mpx_status; the message is
 *     available from mpx_last_error(ctx) until the next call on that ctx;
 *   - "host" entry points take host pointers and are synchronous at return;
 *   - "_dev" entry points take DEVICE pointers (hipMalloc / torch.Tensor.data_ptr)
 *     plus a hipStream_t passed as void* (NULL = the context's own stream) and
 *     only enqueue work: the caller synchronises;
 *   - a context is bound to one device and is not thread-safe;
 *   - chroma vectors are 12 doubles in pitch-class order C, C#, ... B
 *     (chromagram.py:8).
 */
#ifndef MPX_H
#define MPX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPX_ABI_VERSION 6

typedef enum mpx_status {
    MPX_OK = 0,
    MPX_EINVAL = -1,        /* bad argument (the reference raises ValueError / IndexError) */
    MPX_ENOMEM = -2,        /* host or device allocation failed */
    MPX_EHIP = -3,          /* HIP runtime error */
    MPX_EUNSUPPORTED = -4   /* valid request this build has no kernel for */
} mpx_status;

/* mpx_create flags */
#define MPX_FLAG_F32 0x1    /* opt-in fp32 arithmetic (default: fp64, the reference's dtype) */
#define MPX_FLAG_DETERMINISTIC 0x2 /* ESACF: every gaussian fit is finished on the lane that started it (~3 ms per
                                      176 k frames slower).  NOT needed for reproducible results: by default the runaway
                                      fits still open when the work list runs dry are finished by a cooperative kernel
                                      whose arithmetic is bit-identical to the lane kernel's (same summation tree, same
                                      fused multiply-adds), so every run of every entry point returns the same bits
                                      whichever kernel finished which fit.  The flag (or MPX_DETERMINISTIC=1 in the
                                      environment of mpx_create) remains as a cross-check of exactly that. */

typedef struct mpx_ctx mpx_ctx;

int mpx_abi_version(void);
/* 1 when the library was built with -DMPX_DEV_KNOBS (`make dev` -> libmpx_hip_dev.so): only such a build reads the
 * development switches of csrc/ (MPX_SACF_ABLATE, MPX_FIT_MAXFEV, MPX_SACF_PAIR, MPX_FIT_*, MPX_HE_WG, ...) from the
 * environment, some of which change results.  The release library returns 0 and never looks at them; bench.py refuses
 * to report numbers from a library that returns 1. */
int mpx_dev_knobs(void);
int mpx_device_count(void);
/* NULL on failure; mpx_last_error(NULL) then holds the reason.
 * A context owns ONE set of grow-only device workspaces and reduction counters: it is neither thread-safe nor
 * multi-stream.  Everything enqueued on a context -- through its own stream or through the `stream` argument of the
 * *_dev entry points -- must be ordered on one stream at a time: finish (or order behind) the work of the previous
 * stream before handing the same context another one.  For several batches in flight create one context per batch
 * (bench.py: two contexts, two streams). */
mpx_ctx* mpx_create(int device, int flags);
void mpx_destroy(mpx_ctx* ctx);
const char* mpx_last_error(const mpx_ctx* ctx);
/* Block until everything enqueued on the context's stream has finished. */
int mpx_synchronize(mpx_ctx* ctx);
/* The context's hipStream_t (as void*), for callers that want to order their
 * own work after ours. */
void* mpx_stream(mpx_ctx* ctx);
/* How many kernels (groups of launches that belong together) the context has enqueued since it was created; wraps at 2^32.
 * Non-blocking, and the one entry point that may be called from ANOTHER thread while a method call is running on the context:
 * a driver that keeps several contexts busy on one GPU orders their first launches with it -- chord-detection_amd/corpus.py
 * starts Iterative-F0's context once the main context's counter has moved (round 6; the reference has no counterpart: its
 * methods run one after the other, chord_detection/cli.py). */
unsigned mpx_launch_count(const mpx_ctx* ctx);

/* ---- per-context options (ABI 3) ----------------------------------------------
 * Settings of a context that are not kwargs of the reference's constructors: how much device memory one pass may take and
 * which of two equivalent kernels runs.  None of them changes what is computed (tests/ compare every setting with the
 * oracle); they replace what rounds 1-3 read from the environment in development builds only.
 *   MPX_OPT_IF0_WORKSPACE_BYTES  target size, in bytes, of the Iterative-F0 front-end hand-off buffer of ONE pass
 *                                (8 B x channels per sample).  A single clip / stream that needs more runs in time slices
 *                                of whole frames whose filter state is carried over; so does a clip list while one frame of
 *                                every chunk fits (counted in chunks of 262144 samples: a SOFT bound, the planner may cut a
 *                                list into finer chunks and hold up to 16 x as many frames); a list beyond that -- or beyond
 *                                96 GiB whatever the setting -- is cut in halves that run one after the other.
 *                                Default 32 GiB, minimum 64 MiB.
 *   MPX_OPT_HE_KERNEL            MPX_HE_KERNEL_AUTO (default): 4096- and 8192-sample fp64 frames with <= 256 window bins (none of
 *                                them wrapping below bin 0) run on the wave-per-frame kernel (csrc/mpx_he_wave.hpp),
 *                                everything else on the workgroup-per-frame kernel.  An 8192-sample frame is two passes
 *                                over interleaved halves of its samples: of ONE wave when the call's samples fit the L2
 *                                (< 32 MiB), of a PAIR of waves at the same time -- every cache line fetched once instead of
 *                                twice, the same time -- when they are streamed.  MPX_HE_KERNEL_WAVE_PAIRS forces the pairs,
 *                                MPX_HE_KERNEL_WAVE_SERIAL one wave per frame (A/B; same bits; 4096-sample frames are not
 *                                affected).  MPX_HE_KERNEL_WORKGROUP: the workgroup-per-frame kernel for every shape (A/B).
 * mpx_set_option returns MPX_EINVAL for an unknown option or a value out of range. */
#define MPX_OPT_IF0_WORKSPACE_BYTES 1
#define MPX_OPT_HE_KERNEL 2
#define MPX_HE_KERNEL_AUTO 0
#define MPX_HE_KERNEL_WORKGROUP 1
#define MPX_HE_KERNEL_WAVE_PAIRS 2   /* 8192-sample frames: a pair of waves per frame, one per half of the samples, whatever the size of the call */
#define MPX_HE_KERNEL_WAVE_SERIAL 3  /* 8192-sample frames: one wave per frame (two passes), whatever the size of the call */
int mpx_set_option(mpx_ctx* ctx, int option, int64_t value);
int mpx_get_option(mpx_ctx* ctx, int option, int64_t* value);

/* ---- where the samples live -------------------------------------------------
 * `signal` / `signals` of the entry points below (all but the *_dev ones, which run in place) may point to
 *   - memory of the context's device: read IN PLACE (a corpus synthesised or decoded on the GPU need not travel to the
 *     host and back; until round 6 only the framed methods' batch entry points did so, the others copied inside HBM; 16-bit
 *     samples are converted into a buffer of the context).  The buffer must be complete when the call is made -- the kernels
 *     run on the context's stream, not on the producer's -- and must not change until the call returns;
 *   - host memory, pageable or pinned (mpx_host_alloc, hipHostMalloc, torch pin_memory): one asynchronous copy over
 *     PCIe.  Measured for the 33.5 MB headline signal: 45 GB/s from pageable memory (the runtime pins the pages and
 *     DMAs from them), 49 GB/s from pinned memory, against the 63 GB/s of the link.
 * offsets and all outputs are host memory. */
void* mpx_host_alloc(size_t bytes);   /* pinned host memory for input buffers; NULL on failure */
void mpx_host_free(void* p);

/* ---- framing: dsp/frame.py:5-14 ------------------------------------------
 * Number of frames cut from n samples: ceil(n/frame) when hop == frame (the
 * reference's only mode); for hop < frame, frames start every `hop` samples
 * until one reaches the end of the signal.  The tail is zero padded. */
int64_t mpx_num_frames(int64_t n, int frame, int hop);

/* ---- Harmonic Energy (method 2): harmonic_energy.py:14-16 ctor kwargs ----- */
typedef struct mpx_he_params {
    int num_harmonic;   /* default 2 */
    int num_octave;     /* default 2 */
    int num_bins;       /* default 2 */
} mpx_he_params;

/* One signal, host buffers.  replaces harmonic_energy.py:30-73.
 *   signal[n] float32 samples at `fs` Hz; frame = FFT size: a power of two in
 *   1024..16384 (fast path) or any other size from 2 up (chirp-z on the bins below the highest window: one pass while
 *   frame + highest window bin <= 8192, ~6900 samples with the default windows; beyond that the input decimated
 *   by R <= 64, R passes -- ~77 000 samples at 44.1 kHz); hop in [1, frame].
 *   chroma_frames: optional [F,12] per-frame chroma (NULL to skip);
 *   chroma_sum:    [12] sum over frames == compute_pitches() result. */
int mpx_harmonic_energy(mpx_ctx* ctx, const float* signal, int64_t n, int fs,
                        const mpx_he_params* params, int frame, int hop,
                        double* chroma_frames, double* chroma_sum);

/* C clips packed back to back: clip c is signals[offsets[c] .. offsets[c+1]).
 * Every clip is framed on its own (hop == frame semantics per clip, tail zero
 * padded).  chroma_sums: [C,12]. */
int mpx_harmonic_energy_batch(mpx_ctx* ctx, const float* signals, const int64_t* offsets,
                              int num_clips, int fs, const mpx_he_params* params, int frame,
                              int hop, double* chroma_sums);

/* Device-resident variant of mpx_harmonic_energy: d_signal, d_chroma_frames
 * ([F,12], may be NULL) and d_chroma_sum ([12]) are device pointers. */
int mpx_harmonic_energy_dev(mpx_ctx* ctx, const float* d_signal, int64_t n, int fs,
                            const mpx_he_params* params, int frame, int hop,
                            double* d_chroma_frames, double* d_chroma_sum, void* stream);

/* Debug tap: what MultipitchHarmonicEnergy.dft_maxes records (harmonic_energy.py:36,57-65; its only consumer is the
 * reference's plot).  best_ind[F, 12 * num_octave * num_harmonic] int32, window order n x octave x harmonic: the bin k
 * of the FIRST maximum of x_dft[k], k in range(k0, k1), counted like the reference does -- NEGATIVE inside a window that
 * starts below bin 0 (Python's x_dft[k] wraps to the top of the spectrum there; the chroma entry points wrap the same way) --
 * and INT32_MIN for an empty window (None in the reference).  bounds (may be NULL): [windows, 2] the k0, k1 of every
 * window (harmonic_energy.py:50-55).  Host buffers; every frame size runs the chirp-z kernels here (untuned: a plot aid). */
int mpx_harmonic_energy_argmax(mpx_ctx* ctx, const float* signal, int64_t n, int fs,
                               const mpx_he_params* params, int frame, int hop, int32_t* best_ind, int32_t* bounds);

/* ---- note spelling of librosa.hz_to_note (methods 1, 3, 4) ---------------------
 * The reference adds a detected pitch with `chromagram[librosa.hz_to_note(f, octave=False)] += v`
 * (esacf.py:68-69, periodicity.py:107-108, prime_multif0.py:70-72).  What that does to the five sharp pitch
 * classes depends on the librosa the reference runs with (requirements.txt:3 leaves it unpinned):
 *   MPX_NOTES_UNICODE  librosa >= 0.8 spells "C♯".  Chromagram.__getitem__ maps it to "C#" (chromagram.py:21),
 *                      __setitem__ does not (chromagram.py:27-29), so the sum lands in a stray key that
 *                      __add__ (chromagram.py:42-45) never reads: C#, D#, F#, G#, A# are silently dropped.
 *                      This is what the reference computes with a librosa installed today; it is the default.
 *   MPX_NOTES_ASCII    librosa < 0.8 spells "C#": every pitch class accumulates.  This is what the reference's
 *                      README strings (README.md:33-73) and tests/test.py:14-20 expectations were written for. */
#define MPX_NOTES_UNICODE 0
#define MPX_NOTES_ASCII 1

/* ---- ESACF (method 1): esacf.py:17-31 ctor kwargs ------------------------- */
#define MPX_ENHANCE_LIBROSA010 0   /* librosa >= 0.10 time_stretch semantics (default) */
#define MPX_ENHANCE_NOOP 1         /* time_stretch returns nothing: ESACF = clip(SACF, 0) */

typedef struct mpx_esacf_params {
    int n_peaks_elim;      /* default 6 */
    double peak_thresh;    /* default 0.1 */
    int peak_min_dist;     /* default 10 */
    int enhance_mode;      /* MPX_ENHANCE_* */
    int note_names;        /* MPX_NOTES_* (default MPX_NOTES_UNICODE) */
} mpx_esacf_params;

/* One signal, host buffers.  replaces esacf.py:41-91.  frame = ham_samples
 * (any length in [64, 16384]; the reference default int(fs*46.4/1000) is 1023 at
 * 22050 Hz, 2046 at 44100 Hz, 4454 at 96 kHz, 8184 at 176.4 kHz, 8908 at 192 kHz and
 * 16369 at 352.8 kHz; odd lengths above 4096 and everything above 8192 run an untuned
 * kernel; MPX_EUNSUPPORTED above 16384),
 * hop in [1, frame]. */
int mpx_esacf(mpx_ctx* ctx, const float* signal, int64_t n, int fs,
              const mpx_esacf_params* params, int frame, int hop,
              double* chroma_frames, double* chroma_sum);

int mpx_esacf_batch(mpx_ctx* ctx, const float* signals, const int64_t* offsets, int num_clips,
                    int fs, const mpx_esacf_params* params, int frame, int hop,
                    double* chroma_sums);

int mpx_esacf_dev(mpx_ctx* ctx, const float* d_signal, int64_t n, int fs,
                  const mpx_esacf_params* params, int frame, int hop,
                  double* d_chroma_frames, double* d_chroma_sum, void* stream);

/* ---- Prime-multiF0 (method 4): prime_multif0.py:19-26 ctor kwargs --------------
 * replaces prime_multif0.py:41-91.  For each of 12*num_octave*num_harmonic candidate
 * frequencies the signal is cut into frames of int(8/f*fs) samples (357..1348 at 22050 Hz),
 * Hann windowed, |FFT|/sum(window), lower half of the one-sided spectrum, and
 * harmonic_elim_runs rounds of (argmax -> pitch class += peak; zero the bins whose
 * frequency EQUALS 1..harmonic_multiples_elim-1 times the peak frequency).
 * Frame lengths up to 16384 samples are supported (fs <= ~268 kHz with the defaults; above 6553 samples, i.e.
 * ~107 kHz, in decimated chirp-z passes).
 * Two consecutive frames of a candidate share one complex transform (a + i b, separated by conjugate symmetry): a frame
 * carries ~1e-16 of its PARTNER's spectral magnitudes as noise.  Exactly silent frames are special-cased (exact zeros, like
 * the reference); a frame that is merely far quieter than its neighbour (a decay tail 100+ dB down) may pick another
 * arg-max bin than numpy's per-frame FFT would, but what it adds to the clip's chroma is bounded by that noise: the sums
 * match the reference to 1e-5 relative + 1e-13 of the largest bin (tests/test_gpu_prime.py, 80 / 120 / 200 dB). */
typedef struct mpx_prime_params {
    int num_harmonic;              /* default 1 */
    int num_octave;                /* default 2 */
    int harmonic_multiples_elim;   /* default 5 */
    int harmonic_elim_runs;        /* default 2 */
    int note_names;                /* MPX_NOTES_* (default MPX_NOTES_UNICODE) */
} mpx_prime_params;

int mpx_prime_multif0(mpx_ctx* ctx, const float* signal, int64_t n, int fs,
                      const mpx_prime_params* params, double* chroma_sum);

int mpx_prime_multif0_batch(mpx_ctx* ctx, const float* signals, const int64_t* offsets, int num_clips,
                            int fs, const mpx_prime_params* params, double* chroma_sums);

/* Device-resident variant: d_signal is read in place, d_chroma_sum ([12]) is a device buffer, the kernels are enqueued on
 * `stream` (NULL = the context's) -- after the few KB of host-built frame lists have been uploaded and waited for. */
int mpx_prime_multif0_dev(mpx_ctx* ctx, const float* d_signal, int64_t n, int fs, const mpx_prime_params* params,
                          double* d_chroma_sum, void* stream);

/* ---- Iterative F0 (method 3): iterative_f0.py:21-33, periodicity.py:15-28 kwargs ----
 * replaces iterative_f0.py:54-96 + periodicity.py:48-163: 70-channel resonator filterbank over the
 * WHOLE signal (quirk A.1 kept), warped-FIR compression, full-wave rectifier, (y + LP(y, fc))/2,
 * frames of `frame_size` x Hamming zero-padded to 2*frame_size, sum over channels of |FFT|^power, then the iterative
 * period search / harmonic cancellation per frame.  frame_size (iterative_f0.py:25 takes any integer): 1024, 2048, 4096 and
 * the default 8192 run on the tuned power-of-two kernels; any other size in 16 ... 16384 by chirp-z (correct, untuned: such
 * a call runs in one piece, without the time slices of MPX_OPT_IF0_WORKSPACE_BYTES; above 8192 samples -- round 6 -- the
 * 32768-point convolution is four residues of 8192 points); MPX_EUNSUPPORTED above 16384 samples.
 * Long signals are filtered in chunks of up to 262144 samples, each with a zero-state run-in of
 * mpx_iterative_f0_warmup samples (40960 for the defaults: the chain has decayed to fp64 rounding by then), so
 * chunks run in parallel and shard across GPUs. */
typedef struct mpx_if0_params {
    int frame_size;      /* default 8192 */
    double power;        /* default 1.0 */
    int channels;        /* default 70 */
    double zeta0;        /* default 2.3 */
    double zeta1;        /* default 0.39 */
    int max_voices;      /* default 4 (<= 8) */
    double tau_min;      /* default 1/2100 */
    double tau_max;      /* default 1/40 */
    double tau_prec;     /* default 1e-7 */
    int Q;               /* default 20 (<= 32) */
    int M;               /* default 20 (<= 64) */
    double epsilon1;     /* default 20 */
    double epsilon2;     /* default 320 */
    double gamma;        /* default 0.66 */
    int note_names;      /* MPX_NOTES_* (default MPX_NOTES_UNICODE) */
} mpx_if0_params;

int mpx_iterative_f0(mpx_ctx* ctx, const float* signal, int64_t n, int fs, const mpx_if0_params* params,
                     double* chroma_frames /* [F,12] or NULL */, double* chroma_sum /* [12] */);

int mpx_iterative_f0_batch(mpx_ctx* ctx, const float* signals, const int64_t* offsets, int num_clips, int fs,
                           const mpx_if0_params* params, double* chroma_sums /* [C,12] */);

/* Device-resident variant: d_signal is read in place (a stream that lives in HBM is not copied), d_chroma_frames ([F,12],
 * may be NULL) and d_chroma_sum ([12], may be NULL; not both) are device buffers, the kernels are enqueued on `stream`
 * (NULL = the context's) -- after the chunk / frame tables built on the host have been uploaded and waited for. */
int mpx_iterative_f0_dev(mpx_ctx* ctx, const float* d_signal, int64_t n, int fs, const mpx_if0_params* params,
                         double* d_chroma_frames, double* d_chroma_sum, void* stream);

/* Zero-state run-in (samples, a multiple of 8192, >= 16384) after which this parameter set's filter chain has
 * forgotten its start to fp64 rounding: the part of its n^3 rho^n response older than W samples is <= 1e-13 of the whole,
 * e^-u (u^3 + 3u^2 + 6u + 6) / 6 at u = W (1 - rho), for the slowest pole radius rho of the chain (also returned when
 * pole_radius != NULL; 0.99893 and 40960 for the defaults at any sample rate; until round 3 the bound was the absolute
 * rho^W W^3 <= 1e-15: 65536).  The library uses it for its own
 * chunks; a caller that shards one stream over GPUs starts each shard this many samples early (stream.py).
 * The bound is RELATIVE to the chain's whole response: what a chunk or shard boundary leaves behind is <= 1e-13 of the
 * level of the ~0.9 s of signal in front of it, so a quiet frame right after a passage 100 dB louder sees that passage's
 * tail at 1e-13 x its level rather than at fp64 rounding of its own (the sequential reference has no such boundary).
 * MPX_EINVAL when the chain is unstable or would need more than 4 M samples: mpx_iterative_f0* then refuse too. */
int mpx_iterative_f0_warmup(mpx_ctx* ctx, int fs, const mpx_if0_params* params, int64_t* samples, double* pole_radius);

/* Debug tap: summary spectra Ut [F, 2*frame_size] (iterative_f0.py:80-85), host buffers. */
int mpx_iterative_f0_spectra(mpx_ctx* ctx, const float* signal, int64_t n, int fs, const mpx_if0_params* params,
                             double* ut);

/* The period search alone (periodicity.py:48-163, IterativeF0PeriodicityAnalysis.compute -- what the reference keeps in
 * MultipitchIterativeF0.periodicity_estimator, iterative_f0.py:44): spectra[num_frames, bins] summary spectra in HOST memory,
 * bins == 2 * params->frame_size (the window size the estimator was built with); chroma_frames[num_frames, 12] out.
 * Of `params` the period-search fields, frame_size and note_names are read. */
int mpx_iterative_f0_periodicity(mpx_ctx* ctx, const double* spectra, int64_t num_frames, int bins, int fs,
                                 const mpx_if0_params* params, double* chroma_frames);
/* ABI 6: the same search, also returning what periodicity.py:112 returns next to the chromagram --
 * `(self.voicesaliences.copy(), self.voiceperiods.copy())`: saliences[num_frames, max_voices] and
 * periods[num_frames, max_voices] (seconds; 0 for a voice the search did not detect, periodicity.py:56-57,63-67).
 * Either may be NULL. */
int mpx_iterative_f0_periodicity_voices(mpx_ctx* ctx, const double* spectra, int64_t num_frames, int bins, int fs,
                                        const mpx_if0_params* params, double* chroma_frames, double* saliences,
                                        double* periods);

/* ---- PCM_16 input (ABI 6) -------------------------------------------------------
 * The reference's audio is 16-bit PCM: librosa.load (multipitch.py:24-30) hands the methods `int16 / 32768` as float32.
 * These entry points take the int16 samples as the WAV file holds them (host memory, pageable or pinned; or memory of the
 * context's device) and convert x / 32768 ON THE DEVICE -- exact in float32 (16 significant bits, a power-of-two divisor) --
 * so HALF the bytes cross PCIe and the results are bit-equal to the float32 entry points fed `pcm / 32768.0f`
 * (tests/test_gpu_pcm16.py).  Same arguments, outputs and errors as mpx_harmonic_energy / mpx_esacf / mpx_prime_multif0 /
 * mpx_iterative_f0 otherwise.  Use them when the file is mono PCM_16 at the rate the method runs at; anything that has to be
 * resampled or down-mixed goes through the float32 entry points. */
int mpx_harmonic_energy_pcm16(mpx_ctx* ctx, const int16_t* pcm, int64_t n, int fs, const mpx_he_params* params,
                              int frame, int hop, double* chroma_frames, double* chroma_sum);
int mpx_esacf_pcm16(mpx_ctx* ctx, const int16_t* pcm, int64_t n, int fs, const mpx_esacf_params* params, int frame,
                    int hop, double* chroma_frames, double* chroma_sum);
int mpx_prime_multif0_pcm16(mpx_ctx* ctx, const int16_t* pcm, int64_t n, int fs, const mpx_prime_params* params,
                            double* chroma_sum);
int mpx_iterative_f0_pcm16(mpx_ctx* ctx, const int16_t* pcm, int64_t n, int fs, const mpx_if0_params* params,
                           double* chroma_frames /* [F,12] or NULL */, double* chroma_sum /* [12] */);

/* Debug taps for parity tests: per-frame intermediates of the ESACF chain,
 * host buffers, each [F, len]:
 *   MPX_STAGE_WFIR  len = frame            dsp/wfir.py:25-43
 *   MPX_STAGE_XLO   len = frame            esacf.py:51
 *   MPX_STAGE_XHI   len = frame            esacf.py:47-49
 *   MPX_STAGE_SACF  len = (frame-1)/2      esacf.py:93-105
 *   MPX_STAGE_ESACF len = (frame-1)/2      esacf.py:108-129                  */
#define MPX_STAGE_WFIR 0
#define MPX_STAGE_XLO 1
#define MPX_STAGE_XHI 2
#define MPX_STAGE_SACF 3
#define MPX_STAGE_ESACF 4
int mpx_esacf_stage(mpx_ctx* ctx, int stage, const float* signal, int64_t n, int fs,
                    const mpx_esacf_params* params, int frame, int hop, double* out);

/* Warped-FIR design constants (dsp/wfir.py:13-21 calls scipy.signal.remez(13, ...)
 * every frame; here they are host constants).  Tables for fs = 22050 and 44100
 * are built in; for any other sample rate the host registers the 13 taps once. */
int mpx_set_remez_taps(mpx_ctx* ctx, int fs, const double* taps13);

/* Host-callable copy of the device gaussian peak fit (peakutils.interpolate ->
 * scipy curve_fit -> MINPACK lmdif, esacf.py:60) so the restatement can be unit
 * tested without a GPU.  Returns MINPACK's info code (1..4 = converged) or a
 * negative mpx_status; m in [3, 21]. */
int mpx_test_gaussian_fit(const double* xs, const double* ys, int m, double* center);

/* Host-callable copy of the device code for |X|^0.67 (esacf.py:95-101) from x = |X|^2: out[i] = x[i]^(0.67/2) by the table
 * form the SACF kernels use (csrc/mpx_pow067.hpp), so its accuracy can be checked against long-double pow without a GPU. */
int mpx_test_pow067(const double* x, int n, double* out);

/* The quotients and square roots of the device gaussian fit (csrc/mpx_lm.hpp: lm_div, lm_sqrt -- hardware estimate, one
 * Newton step, residual step, instead of the compiler's IEEE sequences), run ON THE GPU so that their rounding can be held
 * against IEEE division / square root by a test (ABI 5): quot[i] = lm_div(a[i], b[i]), root[i] = lm_sqrt(b[i]); host
 * pointers, n >= 0.  Returns MPX_EINVAL for NULL pointers, MPX_ENOMEM / MPX_EHIP when the device refuses. */
int mpx_test_lm_div_sqrt(mpx_ctx* ctx, const double* a, const double* b, int n, double* quot, double* root);

/* ---- timing helper (HIP events on the stream the kernels run on) ----------
 * mpx_timer_begin records an event on `stream` (NULL = context stream);
 * mpx_timer_end records the closing event, waits for it and returns the
 * elapsed milliseconds in *ms. */
int mpx_timer_begin(mpx_ctx* ctx, void* stream);
int mpx_timer_end(mpx_ctx* ctx, void* stream, float* ms);

/* ---- per-kernel timing ------------------------------------------------------
 * Between mpx_profile_begin and mpx_profile_end the library records a HIP event on the launching stream in front of
 * every kernel it launches (and one behind the last kernel of a call); the time from one event to the next is booked
 * on the kernel launched in between.  mpx_profile_end waits for the events and writes one line per kernel name,
 * "name launches total_ms\n", in order of first launch, into report[cap] (NUL terminated).  The events sit between
 * back-to-back launches on one stream, so a figure includes the launch gap (a few microseconds): meant for kernels
 * of a tenth of a millisecond and up -- bench.py times the 50-microsecond Harmonic-Energy kernel over many launches
 * with mpx_timer_begin/_end instead.  Off by default: costs nothing unless enabled. */
int mpx_profile_begin(mpx_ctx* ctx);
int mpx_profile_end(mpx_ctx* ctx, char* report, int cap);
/* ESACF calls made while profiling was on: stats3 = {gaussian peak fits, MINPACK function evaluations they took, fits that
 * were handed to the cooperative kernel}.  (What the fit kernels' roofline is counted in: bench.py.) */
int mpx_esacf_fit_stats(mpx_ctx* ctx, int64_t* stats3);

#ifdef __cplusplus
}
#endif
#endif /* MPX_H */
