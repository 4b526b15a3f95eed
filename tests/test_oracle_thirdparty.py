"""CPU: the restated third-party pieces of the ESACF back half (oracle/thirdparty.py) against independent
implementations that ARE installed here: librosa's STFT / ISTFT conventions against scipy.signal.stft / istft, the
phase vocoder against its rate-1 identity, peakutils.indexes against a brute-force reading of its published rule.
(librosa and peakutils themselves are not installed and cannot be fetched: stages a7-a8 stay "parity unpinned", but
every building block that has an installed counterpart is pinned here.)"""
import numpy as np
import pytest
import scipy.signal

from oracle import thirdparty as tp


def test_stft_matches_scipy_stft():
    """librosa.stft(n_fft=2048, hop=512, periodic Hann, center=True, zero padding) == scipy.signal.stft with
    boundary='zeros', up to scipy's 1/sum(window) scaling."""
    rng = np.random.default_rng(1)
    w = tp.hann_periodic(tp.N_FFT)
    np.testing.assert_allclose(w, scipy.signal.get_window("hann", tp.N_FFT, fftbins=True), atol=1e-15)
    for n in (511, 1022, 2047, 3000, 5000):
        y = np.clip(rng.standard_normal(n), 0, None)       # like a clipped SACF
        D = tp.stft(y)
        # centre=True with zero padding, done by hand: scipy shrinks nperseg for inputs shorter than a window
        ypad = np.concatenate([np.zeros(tp.N_FFT // 2), y, np.zeros(tp.N_FFT // 2)])
        _, _, Z = scipy.signal.stft(ypad, window="hann", nperseg=tp.N_FFT, noverlap=tp.N_FFT - tp.HOP, nfft=tp.N_FFT,
                                    boundary=None, padded=False, return_onesided=True)
        assert D.shape == Z.shape == (tp.N_FFT // 2 + 1, 1 + n // tp.HOP)
        np.testing.assert_allclose(D, Z * w.sum(), rtol=0, atol=1e-11 * np.abs(D).max())


def test_istft_matches_scipy_istft():
    """Overlap-add / window-sum-square normalisation / centre trimming of the restated ISTFT == scipy.signal.istft."""
    rng = np.random.default_rng(2)
    w = tp.hann_periodic(tp.N_FFT)
    for n in (2047, 3000, 5000):
        y = np.clip(rng.standard_normal(n), 0, None)
        D = tp.stft(y)
        mine = tp.istft(D, n)
        _, ref = scipy.signal.istft(D / w.sum(), window="hann", nperseg=tp.N_FFT, noverlap=tp.N_FFT - tp.HOP,
                                    nfft=tp.N_FFT, input_onesided=True, boundary=True)
        m = min(n, ref.shape[0])
        np.testing.assert_allclose(mine[:m], ref[:m], rtol=0, atol=1e-12 * np.abs(y).max())
        # and it inverts the STFT wherever whole windows cover the signal
        np.testing.assert_allclose(mine[:m], y[:m], rtol=0, atol=1e-12 * np.abs(y).max())


def test_phase_vocoder_rate_one_is_identity():
    """rate 1: every output column is |D_t| e^{i(accumulated phase)} and the accumulated phase must be angle(D_t)
    modulo 2 pi -- this pins the phase advance / wrap / accumulate order of the restated vocoder."""
    rng = np.random.default_rng(3)
    y = rng.standard_normal(6000)
    D = tp.stft(y)
    P = tp.phase_vocoder(D, 1.0)
    assert P.shape == D.shape
    np.testing.assert_allclose(P, D, rtol=0, atol=1e-9 * np.abs(D).max())
    np.testing.assert_allclose(tp.time_stretch(y, 1.0), y, rtol=0, atol=1e-9)
    # rate 2 halves the number of frames and keeps the first column; its output length is round(n / 2)
    P2 = tp.phase_vocoder(D, 2.0)
    assert P2.shape[1] == -(-D.shape[1] // 2)
    np.testing.assert_allclose(P2[:, 0], D[:, 0], rtol=0, atol=1e-12 * np.abs(D).max())
    np.testing.assert_allclose(np.abs(P2[:, 1]), np.abs(D[:, 2]), rtol=1e-12)
    assert tp.time_stretch(y, 2.0).shape[0] == 3000


def _brute_force_peaks(y, thres, min_dist):
    """peakutils.indexes read literally, with explicit loops: a peak is a sample above `thres*(max-min)+min` whose
    first difference goes from rising to falling, where the zero runs of the difference (plateaus) count as rising in
    their first half and falling in their second (a leading run takes the slope behind it, a trailing run the slope in
    front of it, an all-flat signal has no peaks); then, highest first, every surviving peak removes the other
    candidates within `min_dist` samples."""
    n = len(y)
    level = thres * (max(y) - min(y)) + min(y)
    dy = [y[i + 1] - y[i] for i in range(n - 1)]
    if all(d == 0 for d in dy):
        return []
    slope = list(dy)
    i = 0
    while i < n - 1:
        if dy[i] != 0:
            i += 1
            continue
        j = i
        while j + 1 < n - 1 and dy[j + 1] == 0:
            j += 1
        run = list(range(i, j + 1))                     # dy is zero on [i, j]
        if i == 0:
            for k in run:
                slope[k] = dy[j + 1]
        elif j == n - 2:
            for k in run:
                slope[k] = dy[i - 1]
        else:
            med = float(np.median(run))
            for k in run:
                slope[k] = dy[i - 1] if k < med else dy[j + 1]
        i = j + 1
    cand = [k for k in range(1, n - 1) if slope[k - 1] > 0 and slope[k] < 0 and y[k] > level]
    if len(cand) > 1 and min_dist > 1:
        alive = {k: True for k in cand}
        for k in sorted(cand, key=lambda q: -y[q]):
            if alive[k]:
                for q in cand:
                    if q != k and abs(q - k) <= min_dist:
                        alive[q] = False
        cand = [k for k in cand if alive[k]]
    return cand


def test_peak_indexes_against_a_brute_force_reading():
    rng = np.random.default_rng(4)
    checked = with_plateaus = 0
    for trial in range(300):
        n = int(rng.integers(5, 400))
        y = rng.standard_normal(n)
        kind = trial % 4
        if kind == 1:                                   # clipped like an enhanced SACF: long exact-zero runs
            y = np.clip(y, 0, None)
        elif kind == 2:                                 # quantised: many interior plateaus of every length
            y = np.round(2 * y) / 2 + 1e-3 * np.arange(n) * (rng.random() < 0.5)
        elif kind == 3:                                 # flat head and tail
            y[:int(rng.integers(1, 4))] = y[0]
            y[-int(rng.integers(1, 4)):] = y[-1]
            y = np.clip(y, -0.3, 0.8)
        # distinct heights among the candidates keep "highest first" unambiguous
        y = y + 1e-9 * rng.permutation(n) * (y > y.min())
        if kind == 1:
            y[y < 1e-8] = 0.0
        for thres, min_dist in ((0.1, 10), (0.3, 1), (0.0, 3), (0.5, 25)):
            got = tp.peak_indexes(y, thres, min_dist).tolist()
            want = _brute_force_peaks([float(v) for v in y], thres, min_dist)
            assert got == want, (trial, thres, min_dist, got, want)
            checked += 1
        with_plateaus += bool(np.any(np.diff(y) == 0))
    assert checked == 1200 and with_plateaus > 100
    assert tp.peak_indexes(np.ones(50), 0.1, 10).size == 0           # all flat
    assert tp.peak_indexes(np.array([0.0, 1.0, 1.0, 1.0, 0.0]), 0.1, 1).tolist() == [2]   # plateau peak: its middle
