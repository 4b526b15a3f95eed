"""Oracle: Iterative-F0 (Klapuri) chroma (reference method 3).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates reference
chord_detection/iterative_f0.py:54-96,171-193 and chord_detection/periodicity.py:48-163 in
float64, keeping every quirk of SURVEY App. A on this path (A.1 swapped fc/fs, A.10 fs/tau with tau in
seconds, A.11 multiply-not-divide salience weight, A.12 mirrored 16384-bin spectrum, A.13 unused voices,
A.18 dropped sharps).  scipy.signal.lfilter/butter are called exactly where the reference calls them
(they are installed both here and on the GPU box); oracle/dsp.py's pure-NumPy lfilter is the fallback.
"""
import math

import numpy as np

from . import dsp
from .esacf import dropped_pitch_classes
from .thirdparty import hz_to_pitch_class

try:  # same call the reference makes; C speed matters for 70 channels x whole-signal filters
    from scipy.signal import lfilter as _lfilter
except Exception:  # pragma: no cover
    _lfilter = dsp.lfilter

HAMMINGWINDOWNORM = [0.0011244659258033, 0.11559343551383, 0.42817348241183, 0.81822361914331, 1.0,
                     0.81822361914331, 0.42817348241183, 0.11559343551383, 0.0011244659258033]


def channel_frequencies(channels=70, zeta0=2.3, zeta1=0.39):
    """iterative_f0.py:37-39."""
    return [229 * (10 ** ((zeta1 * c + zeta0) / 21.4) - 1) for c in range(channels)]


def resonator_coefs(fc, fs):
    """iterative_f0.py:171-186 with the arguments AS DECLARED (x, fc, fs); the call site passes
    (x, fs, fc) -- quirk A.1 -- so callers hand in fc = sample rate, fs = channel frequency."""
    J = 4
    A = np.exp(-(3 / J) * np.pi / (fs * np.sqrt(2 ** (1 / J) - 1)))
    cos_theta1 = (1 + A * A) / (2 * A) * np.cos(2 * np.pi * fc / fs)
    cos_theta2 = (2 * A) / (1 + A * A) * np.cos(2 * np.pi * fc / fs)
    rho1 = (1 / 2) * (1 - A * A)
    rho2 = (1 - A * A) * np.sqrt(1 - cos_theta2 ** 2)
    return ([rho1, 0, -rho1], [1, -A * cos_theta1, A * A]), ([rho2], [1, -A * cos_theta2, A * A])


def auditory_channel(x, fs, fc):
    """iterative_f0.py:58-65 for one channel: filterbank (args swapped), wfir, |.|, (y + LP(y, fc))/2."""
    (b1, a1), (b2, a2) = resonator_coefs(fs, fc)  # A.1: called as _auditory_filterbank(x, fs, fc)
    y = _lfilter(b1, a1, x)
    y = _lfilter(b1, a1, y)
    y = _lfilter(b2, a2, y)
    y = _lfilter(b2, a2, y)
    # wfir with the same lfilter the reference uses
    a = float(dsp.bark_warp_coef(fs))
    ys = [None] * 12
    ys[0] = _lfilter([-a, 1.0], [1.0, -a], y)
    for i in range(1, 12):
        ys[i] = _lfilter([-a, 1.0], [1.0, -a], ys[i - 1])
    c = dsp.warped_remez_coefs(fs, 12)
    x_hat = c[0] * y
    for i in range(12):
        x_hat = x_hat + c[i + 1] * ys[i]
    y = y - x_hat
    y = np.abs(y)  # yc[yc < 0] = -yc[yc < 0]
    b, a_ = dsp.butter2(fc, fs, "low")
    return (y + _lfilter(b, a_, y)) / 2.0


def summary_spectra(x, fs, frame_size=8192, power=1.0, channels=70, zeta0=2.3, zeta1=0.39):
    """iterative_f0.py:57-85 -> Ut [F, 2*frame_size]."""
    x = np.asarray(x, dtype=np.float64)  # lfilter on float32 input computes in float64 already
    F = dsp.num_frames(x.shape[0], frame_size)
    Ut = np.zeros((F, 2 * frame_size))
    w = dsp.hamming_sym(frame_size)
    for fc in channel_frequencies(channels, zeta0, zeta1):
        yc = auditory_channel(x, fs, fc)
        frames = dsp.frame_matrix(yc, frame_size) * w
        padded = np.concatenate([frames, np.zeros_like(frames)], axis=-1)
        Ut += np.abs(np.fft.fft(padded, axis=-1)) ** power
    return Ut


class Periodicity:
    """periodicity.py:14-163 restated (plain Python floats; numpy only for the array)."""

    def __init__(self, fs, window_size, max_voices=4, tau_min=1.0 / 2100.0, tau_max=1.0 / 40.0, tau_prec=0.0000001,
                 Q=20, M=20, epsilon1=20, epsilon2=320, gamma=0.66, note_names="unicode"):
        self.dropped = dropped_pitch_classes(note_names)
        self.fs, self.window_size, self.K = fs, window_size, window_size / fs
        self.max_voices, self.tau_min, self.tau_max, self.tau_prec = max_voices, tau_min, tau_max, tau_prec
        self.Q, self.M, self.epsilon1, self.epsilon2, self.gamma = Q, M, epsilon1, epsilon2, gamma
        self.smax = np.zeros(Q)       # NOT cleared between frames or voices (periodicity.py:43, :114-142)
        self.tau_low = np.zeros(Q)
        self.tau_up = np.zeros(Q)

    def smax_fn(self, q, Ur):
        tau = 0.5 * (self.tau_low[q] + self.tau_up[q])
        deltatau = self.tau_up[q] - self.tau_low[q]
        salience = 0.0
        weight_numerator = self.fs / self.tau_low[q] + self.epsilon1
        for m in range(1, self.M):
            lowk = int(m * self.K / (tau + 0.5 * deltatau) + 0.5)
            highk = int(m * self.K / (tau - 0.5 * deltatau) + 0.5)
            Umax = np.amax(Ur[lowk:highk + 1])
            salience += (m * self.fs / self.tau_up[q] + self.epsilon2) * Umax
        return salience * weight_numerator

    def min_search(self, Ur):
        q = 0
        self.tau_low[0] = self.tau_min
        self.tau_up[0] = self.tau_max
        qbest = 0
        while (self.tau_up[qbest] - self.tau_low[qbest]) > self.tau_prec and q < self.Q - 1:
            q += 1
            self.tau_low[q] = (self.tau_low[qbest] + self.tau_up[qbest]) * 0.5
            self.tau_up[q] = self.tau_up[qbest]
            self.tau_up[qbest] = self.tau_low[q]
            self.smax[q] = self.smax_fn(q, Ur)
            self.smax[qbest] = self.smax_fn(qbest, Ur)
            whichq, maxval = 0, self.smax[0]
            for j in range(1, q + 1):
                if self.smax[j] > maxval:
                    maxval, whichq = self.smax[j], j
            qbest = whichq
        return (self.tau_low[qbest] + self.tau_up[qbest]) * 0.5, self.smax[qbest]

    def compute(self, Uk):
        n = Uk.shape[0]
        voices, cancellation_weight = 0, 1.0
        Ud = np.zeros(n)
        Ur = np.array(Uk)
        sal = np.zeros(self.max_voices)
        per = np.zeros(self.max_voices)
        prevmix = mix = 0.0
        while True:
            tau, best = self.min_search(Ur)
            sal[voices], per[voices] = best, tau
            voices += 1
            mix += best
            test = mix / (math.pow(voices, self.gamma))
            if voices >= self.max_voices or test <= prevmix:
                break
            prevmix = test
            topm = int(tau * (self.fs / self.window_size) * n)
            srovertau = self.fs / tau
            weight = srovertau + self.epsilon1
            for m in range(1, topm):
                partialK = m * self.K / tau + 0.5
                if partialK <= n:
                    Urweight = Ur[int(partialK)]
                    Urweight *= weight / (m * srovertau + self.epsilon2)
                    lowk = max(int(partialK - 4), 0)
                    highk = min(int(partialK + 4), n)
                    for j in range(lowk, highk + 1):
                        Ud[j] += HAMMINGWINDOWNORM[int(j - partialK + 4)] * Urweight
            Ur = np.maximum(Uk - cancellation_weight * Ud, 0)
        chroma = np.zeros(12)
        for i in range(self.max_voices):
            with np.errstate(all="ignore"):
                pitch = np.float64(self.fs) / per[i]
            try:
                pc = hz_to_pitch_class(pitch)
            except OverflowError:  # unused voices: fs/0 = inf (quirk A.13)
                continue
            if pc not in self.dropped:  # quirk A.18
                chroma[pc] += sal[i]
        return chroma, sal, per


def iterative_f0_frames(x, fs, frame_size=8192, power=1.0, channels=70, zeta0=2.3, zeta1=0.39, note_names="unicode"):
    """Per-frame chroma [F,12].  One Periodicity object for the whole clip, like the reference
    (iterative_f0.py:45): its smax[] scratch leaks from frame to frame."""
    Ut = summary_spectra(x, fs, frame_size, power, channels, zeta0, zeta1)
    est = Periodicity(fs, frame_size, note_names=note_names)
    return np.array([est.compute(U)[0] for U in Ut]), Ut


def iterative_f0_compute(x, fs, **kw):
    per, _ = iterative_f0_frames(x, fs, **kw)
    acc = np.zeros(12)
    for f in range(per.shape[0]):
        acc = acc + per[f]
    return acc
