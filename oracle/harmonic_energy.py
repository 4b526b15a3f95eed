"""Oracle: Stark-Plumbley harmonic-energy chroma (reference method 2).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates reference
chord_detection/harmonic_energy.py:31-69 in float64 NumPy, vectorised over
frames (the reference loops frame by frame and bin by bin in Python).
"""
import numpy as np

from . import dsp
from .thirdparty import cqt_frequencies, note_to_hz


def he_windows(fs, frame_size, num_harmonic=2, num_octave=2, num_bins=2):
    """Bin windows [k0, k1) and weights 1/h, reference harmonic_energy.py:33-35,
    46-56.  Order: n (12) x octave x harmonic.  k' uses numpy.round
    (half-to-even); divisor is (fs/4)/N (quirk A.2); octave multiplier is
    `*octave` (quirk A.3)."""
    notes = cqt_frequencies(12, fmin=note_to_hz("C3"))
    divisor_ratio = (fs / 4.0) / frame_size
    k0s, k1s, ws = [], [], []
    for n in range(12):
        for octave in range(1, num_octave + 1):
            for harmonic in range(1, num_harmonic + 1):
                k_prime = np.round((notes[n] * octave * harmonic) / divisor_ratio)
                k0s.append(int(k_prime - num_bins * harmonic))
                k1s.append(int(k_prime + num_bins * harmonic))
                ws.append(1.0 / harmonic)
    return (np.array(k0s).reshape(12, -1), np.array(k1s).reshape(12, -1),
            np.array(ws).reshape(12, -1))


def he_spectrum(frames):
    """sqrt(|rfft(x * hamming_sym(N))|), reference harmonic_energy.py:42-43."""
    frames = np.asarray(frames, dtype=np.float64)
    n = frames.shape[-1]
    return np.sqrt(np.abs(np.fft.rfft(frames * dsp.hamming_sym(n), axis=-1)))


def he_chroma_from_spectrum(x_dft, fs, frame_size, num_harmonic=2, num_octave=2, num_bins=2):
    """reference harmonic_energy.py:44-67: per pitch class, sum over octave and
    harmonic of (max over the half-open bin window) / harmonic."""
    k0, k1, w = he_windows(fs, frame_size, num_harmonic, num_octave, num_bins)
    x_dft = np.atleast_2d(x_dft)
    out = np.zeros((x_dft.shape[0], 12))
    nbins = x_dft.shape[-1]
    for n in range(12):
        for j in range(k0.shape[1]):
            a, b = int(k0[n, j]), int(k1[n, j])
            if b <= a:
                # empty python range: nothing is indexed, the -inf sentinel survives (harmonic_energy.py:49,57)
                out[:, n] += -np.inf * w[n, j]
                continue
            if a < -nbins or b > nbins:
                raise IndexError("harmonic-energy window [%d,%d) outside spectrum of %d bins" % (a, b, nbins))
            if a < 0:
                # the reference indexes x_dft[k] for k in range(k0, k1) (harmonic_energy.py:58-59): a negative k is plain
                # Python indexing and wraps to the top of the spectrum, x_dft[nbins + k]
                idx = [k + nbins if k < 0 else k for k in range(a, b)]
                out[:, n] += x_dft[:, idx].max(axis=-1) * w[n, j]
                continue
            out[:, n] += x_dft[:, a:b].max(axis=-1) * w[n, j]
    return out


def he_argmax(x, fs, frame_size=8192, hop=None, num_harmonic=2, num_octave=2, num_bins=2):
    """best_ind of every window and frame [F, 12 * num_octave * num_harmonic]: the k (as the reference counts it: negative in
    a wrapped window) of the FIRST maximum of x_dft[k], k in range(k0, k1) -- the middle entry of the (k0, best_ind, k1)
    tuples in MultipitchHarmonicEnergy.dft_maxes (harmonic_energy.py:57-65); -1 - nbins for an empty window (None there)."""
    spec = np.atleast_2d(he_spectrum(dsp.frame_matrix(x, frame_size, hop)))
    k0, k1, _ = he_windows(fs, frame_size, num_harmonic, num_octave, num_bins)
    nbins = spec.shape[-1]
    out = np.full((spec.shape[0], k0.size), -1 - nbins, dtype=np.int64)
    for j, (a, b) in enumerate(zip(k0.reshape(-1), k1.reshape(-1))):
        if b <= a:
            continue
        if a < -nbins or b > nbins:
            raise IndexError("harmonic-energy window [%d,%d) outside spectrum of %d bins" % (a, b, nbins))
        idx = [k + nbins if k < 0 else k for k in range(int(a), int(b))]
        out[:, j] = int(a) + np.argmax(spec[:, idx], axis=-1)   # numpy.argmax: first maximum, like the strict > of the loop
    return out


def he_frames(x, fs, frame_size=8192, hop=None, num_harmonic=2, num_octave=2, num_bins=2):
    """Per-frame chroma [F,12] for a 1-D signal."""
    frames = dsp.frame_matrix(x, frame_size, hop)
    return he_chroma_from_spectrum(he_spectrum(frames), fs, frame_size,
                                   num_harmonic, num_octave, num_bins)


def he_compute(x, fs, frame_size=8192, hop=None, **kw):
    """Summed chroma [12] == MultipitchHarmonicEnergy.compute_pitches()
    (reference harmonic_energy.py:30-73); the sum runs frame by frame in
    float64 like Chromagram.__add__ (chromagram.py:42-45)."""
    per = he_frames(x, fs, frame_size, hop, **kw)
    acc = np.zeros(12)
    for f in range(per.shape[0]):
        acc = acc + per[f]
    return acc
