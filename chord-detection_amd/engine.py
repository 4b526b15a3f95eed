"""Thin host wrapper around one mpx_ctx (one per device per process).

Mirrors the reference's error behaviour: bad arguments -> ValueError (the
reference raises ValueError/IndexError from NumPy code), device problems ->
RuntimeError.
"""
import ctypes as C
import threading

import numpy as np

from . import _lib

_engines = {}
_lock = threading.Lock()


class MpxError(RuntimeError):
    pass


def _wait_for_producer(t):
    """The host waits until what torch's CURRENT stream has queued so far on tensor t's device is done.  (As an event recorded
    there and waited for it measured 11-14 us slower per call than this -- he_default_8192 157 -> 171 us -- and changed nothing
    for the corpus driver's contexts, profiles/r6/corpus_first_launch.txt.)"""
    import torch
    torch.cuda.current_stream(t.device).synchronize()


class _DevFlat:
    """A device tensor dressed like the packed numpy array the batch entry points take (`.ctypes.data_as`)."""

    def __init__(self, t):
        self._t = t  # keeps the memory alive for the duration of the call
        self.ctypes = self
        self.shape = (t.numel(),)

    def data_as(self, typ):
        return C.cast(C.c_void_p(self._t.data_ptr()), typ)


class Pcm16:
    """int16 samples as a 16-bit PCM WAV file holds them.  Handed to a single-signal method of the engine they travel to the
    device as they are -- half the PCIe bytes of float32 -- and become x / 32768 there (include/mpx.h "PCM_16 input": exact
    in float32, results bit-equal to the float32 entry points fed `pcm / 32768`).  A plain int16 ndarray is NOT treated this
    way (it is cast to float32 sample values like any other array): the scaling is a property of the file format, not of
    the dtype."""

    def __init__(self, a):
        a = np.asarray(a)
        if a.ndim != 1 or a.dtype != np.int16:
            raise ValueError("Pcm16 takes a 1-D int16 array")
        self.a = np.ascontiguousarray(a)
        self.shape = self.a.shape
        self.ctypes = self.a.ctypes

    def float32(self):
        """what librosa.load / soundfile return for these samples (multipitch.py:24-30)"""
        return self.a.astype(np.float32) / np.float32(32768.0)


class Engine:
    def __init__(self, device=0, f32=False, deterministic=False):
        self.lib = _lib.load()
        self.device = device
        self.f32 = bool(f32)
        self.ctx = self.lib.mpx_create(device, (_lib.MPX_FLAG_F32 if f32 else 0) |
                                       (_lib.MPX_FLAG_DETERMINISTIC if deterministic else 0))
        if not self.ctx:
            msg = self.lib.mpx_last_error(None)
            raise MpxError("mpx_create(device=%d) failed: %s" % (device, (msg or b"?").decode()))

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.mpx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------- helpers
    def _check(self, rc):
        if rc == _lib.MPX_OK:
            return
        msg = (self.lib.mpx_last_error(self.ctx) or b"").decode()
        if rc == _lib.MPX_EINVAL:
            raise ValueError(msg)
        if rc == _lib.MPX_EUNSUPPORTED:
            raise NotImplementedError(msg)
        if rc == _lib.MPX_ENOMEM:
            raise MemoryError(msg)
        raise MpxError(msg)

    @staticmethod
    def _sig(x):
        if isinstance(x, Pcm16):
            return x
        if hasattr(x, "is_cuda") and x.is_cuda:   # 1-D float32 tensor on the device: handed over in place (include/mpx.h)
            import torch
            if x.dim() != 1 or x.dtype != torch.float32 or not x.is_contiguous():
                raise ValueError("a device signal must be a contiguous 1-D float32 tensor")
            _wait_for_producer(x)
            return _DevFlat(x)
        x = np.asarray(x)
        if x.ndim != 1:
            raise ValueError("Only 1D numpy ndarrays are supported")  # dsp/frame.py:6-7
        return np.ascontiguousarray(x, dtype=np.float32)

    @staticmethod
    def _pack(clips):
        if hasattr(clips, "is_cuda") and clips.is_cuda:
            # [C, L] float32 tensor already on the device (e.g. a synthesised corpus chunk): handed over in place, the
            # library reads it there (include/mpx.h, "where the samples live") instead of device -> host -> device
            import torch
            if clips.dim() != 2 or clips.dtype != torch.float32 or not clips.is_contiguous():
                raise ValueError("device clips must be a contiguous float32 [clips, samples] tensor")
            _wait_for_producer(clips)   # complete before the library's stream reads it
            return _DevFlat(clips), np.arange(clips.shape[0] + 1, dtype=np.int64) * clips.shape[1]
        if isinstance(clips, np.ndarray) and clips.ndim == 2:
            # [C, L] array of equal-length clips: already packed back to back, no copy if float32 C-contiguous
            flat = np.ascontiguousarray(clips, dtype=np.float32).reshape(-1)
            return flat, np.arange(clips.shape[0] + 1, dtype=np.int64) * clips.shape[1]
        arrs = [Engine._sig(c) for c in clips]
        if any(isinstance(a, _DevFlat) for a in arrs):
            raise ValueError("a list of device tensors cannot be packed: stack equal-length clips into one [clips, samples] tensor")
        offsets = np.zeros(len(arrs) + 1, dtype=np.int64)
        for i, a in enumerate(arrs):
            offsets[i + 1] = offsets[i] + a.shape[0]
        flat = np.concatenate(arrs) if arrs else np.zeros(0, dtype=np.float32)
        return np.ascontiguousarray(flat, dtype=np.float32), offsets

    def num_frames(self, n, frame, hop=None):
        return int(self.lib.mpx_num_frames(int(n), int(frame), int(hop or frame)))

    def synchronize(self):
        self._check(self.lib.mpx_synchronize(self.ctx))

    @property
    def stream(self):
        return self.lib.mpx_stream(self.ctx)

    def launch_count(self):
        """mpx_launch_count: kernels this context has enqueued so far (non-blocking; callable from another thread while a method
        call runs on the context).  The corpus driver's side threads wait for it to move (corpus._start_side)."""
        return int(self.lib.mpx_launch_count(self.ctx))

    def set_option(self, name, value):
        """mpx_set_option: "if0_workspace_bytes" (cap of one Iterative-F0 pass' hand-off buffer) or "he_kernel"
        (0 auto, 1 workgroup-per-frame kernel for every shape).  Results do not depend on either."""
        if not hasattr(self.lib, "mpx_set_option"):
            raise MpxError("the loaded library (MPX_LIB_PATH, an earlier ABI) has no mpx_set_option")
        self._check(self.lib.mpx_set_option(self.ctx, _lib.OPTIONS[name], int(value)))

    def get_option(self, name):
        if not hasattr(self.lib, "mpx_get_option"):
            raise MpxError("the loaded library (MPX_LIB_PATH, an earlier ABI) has no mpx_get_option")
        v = C.c_int64(0)
        self._check(self.lib.mpx_get_option(self.ctx, _lib.OPTIONS[name], C.byref(v)))
        return int(v.value)

    # ------------------------------------------------------------- method 2
    def harmonic_energy(self, x, fs, frame=8192, hop=None, num_harmonic=2, num_octave=2, num_bins=2,
                        return_frames=False):
        x = self._sig(x)
        hop = int(hop or frame)
        p = _lib.HeParams(num_harmonic, num_octave, num_bins)
        nf = max(self.num_frames(x.shape[0], frame, hop), 0)
        total = np.zeros(12, dtype=np.float64)
        frames = np.zeros((nf, 12), dtype=np.float64) if return_frames else None
        fn, ptr = (self.lib.mpx_harmonic_energy_pcm16, _lib._sp) if isinstance(x, Pcm16) else (self.lib.mpx_harmonic_energy, _lib._fp)
        self._check(fn(
            self.ctx, x.ctypes.data_as(ptr), x.shape[0], int(fs), C.byref(p), int(frame), hop,
            frames.ctypes.data_as(_lib._dp) if return_frames and nf else None, total.ctypes.data_as(_lib._dp)))
        return (total, frames) if return_frames else total

    def harmonic_energy_batch(self, clips, fs, frame=8192, hop=None, num_harmonic=2, num_octave=2, num_bins=2):
        flat, offsets = self._pack(clips)
        p = _lib.HeParams(num_harmonic, num_octave, num_bins)
        out = np.zeros((len(offsets) - 1, 12), dtype=np.float64)
        self._check(self.lib.mpx_harmonic_energy_batch(
            self.ctx, flat.ctypes.data_as(_lib._fp), offsets.ctypes.data_as(_lib._ip), len(offsets) - 1, int(fs),
            C.byref(p), int(frame), int(hop or frame), out.ctypes.data_as(_lib._dp)))
        return out

    def harmonic_energy_argmax(self, x, fs, frame=8192, hop=None, num_harmonic=2, num_octave=2, num_bins=2):
        """[F, 12 * num_octave * num_harmonic] int32: bin of each window's first maximum, counted like the reference does
        (negative inside a wrapped window; INT32_MIN: empty window), and the windows' [k0, k1) bounds [windows, 2] --
        mpx_harmonic_energy_argmax, the dft_maxes tap."""
        x = self._sig(x)
        if isinstance(x, Pcm16):
            x = x.float32()   # (a plot aid: no PCM_16 form of the tap)
        if isinstance(x, _DevFlat):
            raise ValueError("harmonic_energy_argmax takes a host signal")
        hop = int(hop or frame)
        p = _lib.HeParams(num_harmonic, num_octave, num_bins)
        nf = max(self.num_frames(x.shape[0], frame, hop), 0)
        nwin = 12 * int(num_octave) * int(num_harmonic)
        out = np.zeros((nf, nwin), dtype=np.int32)
        bounds = np.zeros((nwin, 2), dtype=np.int32)
        i32 = C.POINTER(C.c_int32)
        self._check(self.lib.mpx_harmonic_energy_argmax(
            self.ctx, x.ctypes.data_as(_lib._fp), x.shape[0], int(fs), C.byref(p), int(frame), hop,
            out.ctypes.data_as(i32), bounds.ctypes.data_as(i32)))
        return out, bounds

    def harmonic_energy_dev(self, d_signal, n, fs, frame, hop, d_frames, d_sum, stream=None,
                            num_harmonic=2, num_octave=2, num_bins=2):
        """Device pointers (ints).  Only enqueues; the caller synchronises."""
        p = _lib.HeParams(num_harmonic, num_octave, num_bins)
        self._check(self.lib.mpx_harmonic_energy_dev(self.ctx, d_signal, int(n), int(fs), C.byref(p), int(frame),
                                                     int(hop), d_frames, d_sum, stream))

    # ------------------------------------------------------------- method 1
    @staticmethod
    def _notes(note_names):
        """'unicode' (librosa >= 0.8: the reference drops the five sharps, include/mpx.h) or 'ascii' (they count)."""
        if note_names not in _lib.NOTE_NAMES:
            raise ValueError("note_names must be one of %s" % sorted(_lib.NOTE_NAMES))
        return _lib.NOTE_NAMES[note_names]

    @staticmethod
    def _esacf_params(n_peaks_elim, peak_thresh, peak_min_dist, enhance_mode, note_names="unicode"):
        modes = {"librosa010": _lib.MPX_ENHANCE_LIBROSA010, "noop": _lib.MPX_ENHANCE_NOOP}
        if enhance_mode not in modes:
            raise ValueError("enhance_mode must be one of %s" % sorted(modes))
        return _lib.EsacfParams(int(n_peaks_elim), float(peak_thresh), int(peak_min_dist), modes[enhance_mode],
                                Engine._notes(note_names))

    def _ensure_remez(self, fs):
        """Warped-FIR taps are a design constant (dsp/wfir.py:13-21); 22050/44100 Hz are built in, other
        rates are designed once on the host exactly as the reference does (scipy.signal.remez)."""
        fs = int(fs)
        if fs in (22050, 44100) or fs in getattr(self, "_remez_done", set()):
            return
        import scipy.signal
        r = min(20000, fs / 2 - 1)
        taps = np.ascontiguousarray(scipy.signal.remez(13, [0, 19, 20, r, r + 1, 0.5 * fs], [0, 1, 0], fs=fs),
                                    dtype=np.float64)
        self._check(self.lib.mpx_set_remez_taps(self.ctx, fs, taps.ctypes.data_as(_lib._dp)))
        self._remez_done = getattr(self, "_remez_done", set()) | {fs}

    def esacf(self, x, fs, frame, hop=None, n_peaks_elim=6, peak_thresh=0.1, peak_min_dist=10,
              enhance_mode="librosa010", return_frames=False, note_names="unicode"):
        x = self._sig(x)
        hop = int(hop or frame)
        p = self._esacf_params(n_peaks_elim, peak_thresh, peak_min_dist, enhance_mode, note_names)
        self._ensure_remez(fs)
        nf = max(self.num_frames(x.shape[0], frame, hop), 0)
        total = np.zeros(12, dtype=np.float64)
        frames = np.zeros((nf, 12), dtype=np.float64) if return_frames else None
        fn, ptr = (self.lib.mpx_esacf_pcm16, _lib._sp) if isinstance(x, Pcm16) else (self.lib.mpx_esacf, _lib._fp)
        self._check(fn(
            self.ctx, x.ctypes.data_as(ptr), x.shape[0], int(fs), C.byref(p), int(frame), hop,
            frames.ctypes.data_as(_lib._dp) if return_frames and nf else None, total.ctypes.data_as(_lib._dp)))
        return (total, frames) if return_frames else total

    def esacf_batch(self, clips, fs, frame, hop=None, n_peaks_elim=6, peak_thresh=0.1, peak_min_dist=10,
                    enhance_mode="librosa010", note_names="unicode"):
        flat, offsets = self._pack(clips)
        p = self._esacf_params(n_peaks_elim, peak_thresh, peak_min_dist, enhance_mode, note_names)
        self._ensure_remez(fs)
        out = np.zeros((len(offsets) - 1, 12), dtype=np.float64)
        self._check(self.lib.mpx_esacf_batch(
            self.ctx, flat.ctypes.data_as(_lib._fp), offsets.ctypes.data_as(_lib._ip), len(offsets) - 1, int(fs),
            C.byref(p), int(frame), int(hop or frame), out.ctypes.data_as(_lib._dp)))
        return out

    def esacf_dev(self, d_signal, n, fs, frame, hop, d_frames, d_sum, stream=None, n_peaks_elim=6,
                  peak_thresh=0.1, peak_min_dist=10, enhance_mode="librosa010", note_names="unicode"):
        p = self._esacf_params(n_peaks_elim, peak_thresh, peak_min_dist, enhance_mode, note_names)
        self._ensure_remez(fs)
        self._check(self.lib.mpx_esacf_dev(self.ctx, d_signal, int(n), int(fs), C.byref(p), int(frame), int(hop),
                                           d_frames, d_sum, stream))

    def esacf_stage(self, stage, x, fs, frame, hop=None, n_peaks_elim=6, peak_thresh=0.1, peak_min_dist=10,
                    enhance_mode="librosa010"):
        """Per-frame intermediates [F, len] for parity tests (wfir, x_lo, x_hi, sacf, esacf)."""
        x = self._sig(x)
        hop = int(hop or frame)
        p = self._esacf_params(n_peaks_elim, peak_thresh, peak_min_dist, enhance_mode)
        self._ensure_remez(fs)
        nf = max(self.num_frames(x.shape[0], frame, hop), 0)
        sid = _lib.STAGES[stage]
        length = frame if sid <= 2 else (frame - 1) // 2
        out = np.zeros((nf, length), dtype=np.float64)
        self._check(self.lib.mpx_esacf_stage(self.ctx, sid, x.ctypes.data_as(_lib._fp), x.shape[0], int(fs),
                                             C.byref(p), int(frame), hop, out.ctypes.data_as(_lib._dp)))
        return out

    # ------------------------------------------------------------- method 3
    @staticmethod
    def _if0_params(frame_size=8192, power=1.0, channels=70, zeta0=2.3, zeta1=0.39, max_voices=4,
                    tau_min=1.0 / 2100.0, tau_max=1.0 / 40.0, tau_prec=0.0000001, Q=20, M=20, epsilon1=20,
                    epsilon2=320, gamma=0.66, note_names="unicode"):
        return _lib.If0Params(int(frame_size), float(power), int(channels), float(zeta0), float(zeta1),
                              int(max_voices), float(tau_min), float(tau_max), float(tau_prec), int(Q), int(M),
                              float(epsilon1), float(epsilon2), float(gamma), Engine._notes(note_names))

    def iterative_f0(self, x, fs, return_frames=False, **kw):
        x = self._sig(x)
        p = self._if0_params(**kw)
        self._ensure_remez(fs)
        nf = max(self.num_frames(x.shape[0], p.frame_size, p.frame_size), 0)
        if isinstance(x, _DevFlat) and nf:
            # a signal that lives in HBM is read IN PLACE (mpx_iterative_f0_dev): the host entry point would copy it into a
            # workspace of its own first -- 635 MB for an hour of audio, allocated inside the first call of a process
            import torch
            t = x._t
            d_frames = torch.empty((nf, 12), dtype=torch.float64, device=t.device)
            d_sum = torch.empty(12, dtype=torch.float64, device=t.device)
            self._check(self.lib.mpx_iterative_f0_dev(self.ctx, t.data_ptr(), t.numel(), int(fs), C.byref(p),
                                                      d_frames.data_ptr(), d_sum.data_ptr(), None))
            self.synchronize()
            total = d_sum.cpu().numpy()
            return (total, d_frames.cpu().numpy()) if return_frames else total
        total = np.zeros(12, dtype=np.float64)
        frames = np.zeros((nf, 12), dtype=np.float64) if return_frames else None
        fn, ptr = (self.lib.mpx_iterative_f0_pcm16, _lib._sp) if isinstance(x, Pcm16) else (self.lib.mpx_iterative_f0, _lib._fp)
        self._check(fn(
            self.ctx, x.ctypes.data_as(ptr), x.shape[0], int(fs), C.byref(p),
            frames.ctypes.data_as(_lib._dp) if return_frames and nf else None, total.ctypes.data_as(_lib._dp)))
        return (total, frames) if return_frames else total

    def iterative_f0_periodicity(self, spectra, fs, return_voices=False, **kw):
        """The period search alone on summary spectra [F, 2 * frame_size] (or one row): per-frame chroma [F, 12] --
        mpx_iterative_f0_periodicity (periodicity.py:48-163)."""
        u = np.ascontiguousarray(np.atleast_2d(np.asarray(spectra, dtype=np.float64)))
        if u.ndim != 2:
            raise ValueError("spectra must be [frames, 2 * frame_size]")
        kw.setdefault("frame_size", u.shape[1] // 2)
        p = self._if0_params(**kw)
        out = np.zeros((u.shape[0], 12), dtype=np.float64)
        if return_voices:   # periodicity.py:112: (voicesaliences, voiceperiods), one row of max_voices per frame
            sal = np.zeros((u.shape[0], p.max_voices), dtype=np.float64)
            per = np.zeros((u.shape[0], p.max_voices), dtype=np.float64)
            self._check(self.lib.mpx_iterative_f0_periodicity_voices(
                self.ctx, u.ctypes.data_as(_lib._dp), u.shape[0], u.shape[1], int(fs), C.byref(p), out.ctypes.data_as(_lib._dp),
                sal.ctypes.data_as(_lib._dp), per.ctypes.data_as(_lib._dp)))
            return out, sal, per
        self._check(self.lib.mpx_iterative_f0_periodicity(self.ctx, u.ctypes.data_as(_lib._dp), u.shape[0], u.shape[1], int(fs),
                                                          C.byref(p), out.ctypes.data_as(_lib._dp)))
        return out

    def iterative_f0_batch(self, clips, fs, **kw):
        flat, offsets = self._pack(clips)
        p = self._if0_params(**kw)
        self._ensure_remez(fs)
        out = np.zeros((len(offsets) - 1, 12), dtype=np.float64)
        self._check(self.lib.mpx_iterative_f0_batch(
            self.ctx, flat.ctypes.data_as(_lib._fp), offsets.ctypes.data_as(_lib._ip), len(offsets) - 1, int(fs),
            C.byref(p), out.ctypes.data_as(_lib._dp)))
        return out

    def iterative_f0_dev(self, d_signal, n, fs, d_frames, d_sum, stream=None, **kw):
        """Device pointers (ints); the signal is read in place.  Only enqueues; the caller synchronises."""
        p = self._if0_params(**kw)
        self._ensure_remez(fs)
        self._check(self.lib.mpx_iterative_f0_dev(self.ctx, d_signal, int(n), int(fs), C.byref(p), d_frames, d_sum, stream))

    def prime_multif0_dev(self, d_signal, n, fs, d_sum, stream=None, num_harmonic=1, num_octave=2, harmonic_multiples_elim=5,
                          harmonic_elim_runs=2, note_names="unicode"):
        p = _lib.PrimeParams(num_harmonic, num_octave, harmonic_multiples_elim, harmonic_elim_runs, self._notes(note_names))
        self._check(self.lib.mpx_prime_multif0_dev(self.ctx, d_signal, int(n), int(fs), C.byref(p), d_sum, stream))

    def iterative_f0_warmup(self, fs, **kw):
        """(run-in samples, slowest pole radius) of this parameter set's filter chain (include/mpx.h); ValueError when the
        chain decays too slowly to be cut into chunks or time shards."""
        p = self._if0_params(**kw)
        w, rho = C.c_int64(0), C.c_double(0.0)
        self._check(self.lib.mpx_iterative_f0_warmup(self.ctx, int(fs), C.byref(p), C.byref(w), C.byref(rho)))
        return int(w.value), float(rho.value)

    def iterative_f0_spectra(self, x, fs, **kw):
        """Summary spectra Ut [F, 2*frame_size] (iterative_f0.py:80-85), for parity tests."""
        x = self._sig(x)
        p = self._if0_params(**kw)
        self._ensure_remez(fs)
        nf = max(self.num_frames(x.shape[0], p.frame_size, p.frame_size), 0)
        ut = np.zeros((nf, 2 * p.frame_size), dtype=np.float64)
        if nf:
            self._check(self.lib.mpx_iterative_f0_spectra(self.ctx, x.ctypes.data_as(_lib._fp), x.shape[0], int(fs),
                                                          C.byref(p), ut.ctypes.data_as(_lib._dp)))
        return ut

    # ------------------------------------------------------------- method 4
    def prime_multif0(self, x, fs, num_harmonic=1, num_octave=2, harmonic_multiples_elim=5, harmonic_elim_runs=2,
                      note_names="unicode"):
        x = self._sig(x)
        p = _lib.PrimeParams(num_harmonic, num_octave, harmonic_multiples_elim, harmonic_elim_runs,
                             self._notes(note_names))
        total = np.zeros(12, dtype=np.float64)
        fn, ptr = (self.lib.mpx_prime_multif0_pcm16, _lib._sp) if isinstance(x, Pcm16) else (self.lib.mpx_prime_multif0, _lib._fp)
        self._check(fn(self.ctx, x.ctypes.data_as(ptr), x.shape[0], int(fs), C.byref(p), total.ctypes.data_as(_lib._dp)))
        return total

    def prime_multif0_batch(self, clips, fs, num_harmonic=1, num_octave=2, harmonic_multiples_elim=5,
                            harmonic_elim_runs=2, note_names="unicode"):
        flat, offsets = self._pack(clips)
        p = _lib.PrimeParams(num_harmonic, num_octave, harmonic_multiples_elim, harmonic_elim_runs,
                             self._notes(note_names))
        out = np.zeros((len(offsets) - 1, 12), dtype=np.float64)
        self._check(self.lib.mpx_prime_multif0_batch(
            self.ctx, flat.ctypes.data_as(_lib._fp), offsets.ctypes.data_as(_lib._ip), len(offsets) - 1, int(fs),
            C.byref(p), out.ctypes.data_as(_lib._dp)))
        return out

    # ------------------------------------------------------------- timing
    def timer_begin(self, stream=None):
        self._check(self.lib.mpx_timer_begin(self.ctx, stream))

    def timer_end(self, stream=None):
        ms = C.c_float(0.0)
        self._check(self.lib.mpx_timer_end(self.ctx, stream, C.byref(ms)))
        return float(ms.value)


    def profile_begin(self):
        """Per-kernel HIP-event timing of everything this context launches until profile_end (include/mpx.h)."""
        self._check(self.lib.mpx_profile_begin(self.ctx))

    def profile_end(self):
        """{kernel name: (launches, total milliseconds)} in order of first launch."""
        buf = C.create_string_buffer(1 << 16)
        self._check(self.lib.mpx_profile_end(self.ctx, buf, len(buf)))
        out = {}
        for line in buf.value.decode().splitlines():
            name, launches, ms = line.rsplit(" ", 2)
            out[name] = (int(launches), float(ms))
        return out


def _fit_stats(self):
    """{fits, evaluations, parked} of the ESACF calls made between profile_begin and profile_end (include/mpx.h)."""
    out = (C.c_int64 * 3)()
    self._check(self.lib.mpx_esacf_fit_stats(self.ctx, out))
    return {"fits": int(out[0]), "evaluations": int(out[1]), "parked": int(out[2])}


Engine.esacf_fit_stats = _fit_stats


class _Pinned:
    def __init__(self, lib, ptr):
        self.lib, self.ptr = lib, ptr

    def __del__(self):
        try:
            self.lib.mpx_host_free(self.ptr)
        except Exception:
            pass


def pinned_empty(n, dtype=np.float32, lib=None):
    """A 1-D NumPy array in pinned host memory (mpx_host_alloc): as an input of the host entry points it crosses PCIe
    in one DMA from the caller's pages (include/mpx.h, "where the samples live").

    Lifetime: the memory is freed (mpx_host_free) when the LAST array that looks at it dies.  The object that frees it hangs
    on the ctypes buffer NumPy keeps as the base of every view -- plain-ndarray views (np.asarray, slices, .view(np.ndarray))
    included -- not on an attribute of the returned array, which such views do not carry."""
    lib = lib or _lib.load()
    dtype = np.dtype(dtype)
    nbytes = int(n) * dtype.itemsize
    ptr = lib.mpx_host_alloc(nbytes)
    if not ptr:
        raise MemoryError("mpx_host_alloc(%d) failed" % nbytes)
    buf = (C.c_char * nbytes).from_address(ptr)
    buf._mpx_owner = _Pinned(lib, ptr)   # dies with the buffer, i.e. after every array built on it
    return np.frombuffer(buf, dtype=dtype, count=int(n))


def get_engine(device=0, f32=False, deterministic=False):
    """Process-wide engine per (device, dtype, lane-mode fits).  `deterministic` = MPX_FLAG_DETERMINISTIC: the slower way
    of computing the SAME bits (include/mpx.h); results are reproducible either way."""
    key = (int(device), bool(f32), bool(deterministic))
    with _lock:
        eng = _engines.get(key)
        if eng is None:
            eng = _engines[key] = Engine(device, f32, deterministic)
        return eng


def device_count():
    return int(_lib.load().mpx_device_count())
