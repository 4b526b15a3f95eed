"""Oracle: ESACF (Tolonen-Karjalainen) chroma (reference method 1).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates reference
chord_detection/esacf.py:41-129 in float64 NumPy.  Stages a4-a6 (wfir, band
split, SACF) follow reference code that runs in the authoring container and are
pinned by fixtures; stages a7-a8 (time_stretch enhancement, peak pick + fit)
call third-party code that is absent -> restated in oracle/thirdparty.py,
PARITY UNPINNED.
"""
import numpy as np

from . import dsp
from . import thirdparty as tp

SACF_K = 0.67  # esacf.py:95-96: self.k is never forwarded (quirk A.6)

# Quirk A.18 (found while pinning the oracle against the reference's code):
# librosa.hz_to_note returns unicode sharps ("C♯"); Chromagram.__getitem__ maps
# '♯'->'#' (chromagram.py:21) but __setitem__ does not (chromagram.py:29), so
# `chromagram[note] += v` (esacf.py:69) stores the sum under a NEW key "C♯" that
# Chromagram.__add__ (chromagram.py:43-44) never reads: every sharp pitch class
# is silently dropped by the hz_to_note-based methods (1, 3, 4).  Method 2
# indexes by int and is unaffected.
# That is the behaviour with librosa >= 0.8 (note_names="unicode", what an install
# of the reference gets today: requirements.txt:3 is unpinned).  librosa < 0.8
# spelled "C#" (note_names="ascii"): nothing is lost, and that is what the
# reference's README strings and tests/test.py:14-20 expectations were written for.
SHARP_PITCH_CLASSES = (1, 3, 6, 8, 10)
NOTE_NAME_MODES = ("unicode", "ascii")


def dropped_pitch_classes(note_names="unicode"):
    """Pitch classes `chromagram[hz_to_note(f)] += v` loses (chromagram.py:19-29)."""
    if note_names not in NOTE_NAME_MODES:
        raise ValueError("note_names must be one of %s" % (NOTE_NAME_MODES,))
    return SHARP_PITCH_CLASSES if note_names == "unicode" else ()


def ham_samples(fs, ham_ms=46.4):
    """reference esacf.py:27."""
    return int(fs * ham_ms / 1000.0)


def band_split(frames, fs):
    """reference esacf.py:45-51 on [F,N] float64 frames -> (x_w, x_lo, x_hi)."""
    x = dsp.wfir(frames, fs, 12)
    x_hi = dsp.highpass_filter(x, fs)
    x_hi = np.clip(x_hi, 0, None)
    x_hi = dsp.lowpass_filter(x_hi, fs, 1000)
    x_lo = dsp.lowpass_filter(x, fs, 1000)
    return x, x_lo, x_hi


def sacf(x_lo, x_hi):
    """reference esacf.py:93-105: real(ifft(sum_ch |fft(x_ch)|**0.67))[:(N-1)//2];
    circular, length-N complex FFT, no window, no zero padding."""
    x_lo = np.asarray(x_lo, dtype=np.float64)
    n = x_lo.shape[-1]
    s = np.abs(np.fft.fft(x_lo, axis=-1)) ** SACF_K + np.abs(np.fft.fft(x_hi, axis=-1)) ** SACF_K
    return np.real(np.fft.ifft(s, axis=-1))[..., :int((n - 1) / 2)]


def esacf_enhance(x2, n_peaks=6, mode="librosa010"):
    """reference esacf.py:108-129.  mode 'librosa010': librosa>=0.10 time_stretch
    semantics (oracle/thirdparty.py); 'noop': time_stretch returns an empty
    array (older librosa on a one-frame STFT; matches the README figure)."""
    x2tmp = np.array(x2, dtype=np.float64)
    m = x2tmp.shape[0]
    for timescale in range(2, n_peaks + 1):
        x2tmp = np.clip(x2tmp, 0, None)
        if mode == "noop":
            stretched = np.zeros(0)
        elif tp.time_stretch_is_truncation(m):
            stretched = x2tmp[:int(round(m / timescale))].copy()
        else:
            stretched = tp.time_stretch(x2tmp, timescale)
        s = np.zeros(m)  # ndarray.resize(x2tmp.shape): zero-extend / truncate
        k = min(m, stretched.shape[0])
        s[:k] = stretched[:k]
        x2tmp = x2tmp - s
        x2tmp = np.clip(x2tmp, 0, None)
    return x2tmp


def frame_chroma(x_esacf, fs, peak_thresh=0.1, peak_min_dist=10, detail=False, note_names="unicode"):
    """reference esacf.py:56-71: peak pick, gaussian interpolation, pitch-class
    scatter.  The i-th *interpolated* lag is paired with the i-th peak index
    (quirk A.8)."""
    peaks = tp.peak_indexes(x_esacf, thres=peak_thresh, min_dist=peak_min_dist)
    interp = tp.peak_interpolate(np.arange(x_esacf.shape[0]), x_esacf, peaks)
    chroma = np.zeros(12)
    dropped = dropped_pitch_classes(note_names)
    for i, tau in enumerate(interp):
        with np.errstate(all="ignore"):
            pitch = fs / tau
        try:
            pc = tp.hz_to_pitch_class(pitch)
        except ValueError:
            continue
        if pc in dropped:
            continue
        chroma[pc] += x_esacf[peaks[i]]
    if detail:
        return chroma, peaks, interp
    return chroma


def frame_fragility(x_esacf, fs, peak_thresh=0.1, peak_min_dist=10, eps=1e-12, trials=2, note_names="ascii"):
    """Test helper: True when the REFERENCE ALGORITHM ITSELF is ill-conditioned on this frame,
    i.e. a relative perturbation of 1e-12 of the ESACF row (far below anything float64 FFTs
    can promise) changes the frame's chroma by more than 1e-6.  That happens when a gaussian
    fit runs away from its 21-sample window (MINPACK then stops wherever its tolerances say,
    and the "centre" decides the pitch class) or a pitch rounds on a pitch-class boundary.
    Such frames are compared on identical inputs only (tests/test_gpu_esacf.py)."""
    base = frame_chroma(x_esacf, fs, peak_thresh, peak_min_dist, note_names=note_names)
    rng = np.random.default_rng(12345)
    for _ in range(trials):
        pert = x_esacf * (1.0 + eps * rng.standard_normal(x_esacf.shape[0]))
        other = frame_chroma(pert, fs, peak_thresh, peak_min_dist, note_names=note_names)
        if not np.allclose(base, other, rtol=1e-6, atol=1e-12):
            return True
    return False


def frame_has_runaway_fit(x_esacf, fs, peak_thresh=0.1, peak_min_dist=10, width=10):
    """Test helper: True when an ACCEPTED gaussian fit of this frame put its centre outside the 21-sample window
    it was fitted on (MINPACK "converged" somewhere along a flat valley: the reference keeps such a centre, and its
    value -- hence the pitch class -- moves with the last bits of the arithmetic even when a 1e-12 perturbation of
    the input happens not to flip it), or when a fit failed, which shifts the index/lag pairing (quirk A.8)."""
    _, peaks, interp = frame_chroma(x_esacf, fs, peak_thresh, peak_min_dist, detail=True)
    if len(interp) != len(peaks):
        return True
    return bool(np.any(np.abs(np.asarray(interp) - np.asarray(peaks, dtype=np.float64)) > width + 0.5))


def runaway_fit_bins(x_esacf, fs, peak_thresh=0.1, peak_min_dist=10, width=10):
    """Test helper.  (pairing_shifted, bins): `pairing_shifted` is True when a gaussian fit of this frame FAILED, which
    shifts the index/lag pairing of every later peak (quirk A.8: then any bin may move with the last bits of the
    arithmetic); `bins` are the pitch classes that an ACCEPTED fit whose centre left its 21-sample window feeds -- such a
    centre sits somewhere along a flat valley of MINPACK's objective, and another rounding of the same algorithm moves
    that peak's height from this bin to another one (so at most 2 bins of the frame change per such fit)."""
    _, peaks, interp = frame_chroma(x_esacf, fs, peak_thresh, peak_min_dist, detail=True, note_names="ascii")
    if len(interp) != len(peaks):
        return True, []
    bins = []
    for pk, tau in zip(peaks, interp):
        if abs(float(tau) - float(pk)) > width + 0.5:
            try:
                with np.errstate(all="ignore"):
                    bins.append(tp.hz_to_pitch_class(fs / tau))
            except (ValueError, OverflowError):
                bins.append(-1)
    return False, bins


def esacf_frames(x, fs, frame_size=None, hop=None, n_peaks_elim=6, peak_thresh=0.1,
                 peak_min_dist=10, enhance_mode="librosa010", note_names="unicode"):
    if frame_size is None:
        frame_size = ham_samples(fs)
    frames = dsp.frame_matrix(x, frame_size, hop)
    _, x_lo, x_hi = band_split(frames, fs)
    s = sacf(x_lo, x_hi)
    out = np.zeros((frames.shape[0], 12))
    for f in range(frames.shape[0]):
        e = esacf_enhance(s[f], n_peaks_elim, enhance_mode)
        out[f] = frame_chroma(e, fs, peak_thresh, peak_min_dist, note_names=note_names)
    return out


def esacf_compute(x, fs, **kw):
    """Summed chroma [12] == MultipitchESACF.compute_pitches() (esacf.py:41-91)."""
    per = esacf_frames(x, fs, **kw)
    acc = np.zeros(12)
    for f in range(per.shape[0]):
        acc = acc + per[f]
    return acc
