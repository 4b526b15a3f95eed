"""Oracle: Prime-multiF0 chroma (reference method 4).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates reference
chord_detection/prime_multif0.py:41-91 in float64 NumPy.  Third-party pieces:
matplotlib.mlab.magnitude_spectrum (installed in the authoring container, so the
fixtures pin it: |fft(x*window)|[:N//2+1] / window.sum(), freqs = fftfreq(N, 1/Fs)),
numpy.hanning, and librosa's closed-form note helpers.
"""
import numpy as np

from . import dsp
from .esacf import dropped_pitch_classes
from .thirdparty import cqt_frequencies, hz_to_pitch_class, note_to_hz


def candidates(fs, num_harmonic=1, num_octave=2):
    """(f_candidate, window_size) in the reference's loop order (prime_multif0.py:49-53)."""
    notes = cqt_frequencies(12, fmin=note_to_hz("C3"))
    out = []
    for n in range(12):
        for octave in range(1, num_octave + 1):
            for harmonic in range(1, num_harmonic + 1):
                f = notes[n] * octave * harmonic
                out.append((f, int((8 / f) * fs)))
    return out


def freq_axis(n, fs):
    """numpy.fft.fftfreq(n, 1/Fs)[:n//2+1] as mlab builds it (exact float ops matter: the
    harmonic elimination compares frequencies with ==, prime_multif0.py:80-81)."""
    val = 1.0 / (n * (1 / fs))
    return np.arange(0, n // 2 + 1) * val


def frame_contributions(s, f, harmonic_multiples_elim=5, harmonic_elim_runs=2, note_names="unicode"):
    """prime_multif0.py:66-82 on one half-spectrum: list of (pitch_class, value)."""
    s = s.copy()
    out = []
    dropped = dropped_pitch_classes(note_names)
    for _ in range(harmonic_elim_runs):
        idx = int(s.argmax(axis=0))
        max_f = f[idx]
        try:
            with np.errstate(all="ignore"):
                pc = hz_to_pitch_class(max_f)
        except (ValueError, OverflowError):
            continue
        if pc not in dropped:  # quirk A.18 (the elimination below still happens)
            out.append((pc, float(s[idx])))
        for k in range(1, harmonic_multiples_elim):
            s[np.where(f == k * max_f)] = 0.0
    return out


def prime_compute(x, fs, num_harmonic=1, num_octave=2, harmonic_multiples_elim=5, harmonic_elim_runs=2,
                  note_names="unicode"):
    """Summed chroma [12] == MultipitchPrimeMultiF0.compute_pitches()."""
    overall = np.zeros(12)
    for _, ws in candidates(fs, num_harmonic, num_octave):
        frames = dsp.frame_matrix(x, ws)
        window = np.hanning(ws)
        spec = np.abs(np.fft.fft(frames * window, axis=-1))[:, :ws // 2 + 1] / window.sum()
        freqs = freq_axis(ws, fs)
        half_s, half_f = int(spec.shape[1] / 2), int(freqs.shape[0] / 2)
        chroma = np.zeros(12)
        for fr in range(frames.shape[0]):
            for pc, v in frame_contributions(spec[fr, :half_s], freqs[:half_f], harmonic_multiples_elim,
                                             harmonic_elim_runs, note_names):
                chroma[pc] += v
        overall = overall + chroma
    return overall
