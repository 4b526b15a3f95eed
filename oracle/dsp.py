"""Oracle: framing, warped-FIR pre-whitening, Butterworth band split.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  float64 NumPy restatement
of reference chord_detection/dsp/{frame,wfir,lowpass}.py and esacf.py:132-134.
The SciPy calls the reference makes (butter, lfilter, remez, hamming) are
replaced by closed forms / tables so that nothing but NumPy is needed; the
golden fixtures were produced by the reference's own code calling real SciPy.
"""
import math

import numpy as np

# scipy.signal.remez(13, [0,19,20,r,r+1,fs/2], [0,1,0], fs=fs), r=min(20000, fs/2-1)
# as called by reference dsp/wfir.py:13-21 (order=12), measured with SciPy 1.15.3
# in the authoring container; fixture G7 in tests/golden/constants.json pins them.
REMEZ_TAPS = {
    22050: [-0.2503141758465685, -0.00010985253437014219, 2.5089273607721457e-05,
            -0.0002589029075222507, 0.00020302128904025308, 0.0002957470030712228,
            1.0000257474401701, 0.0002957470030712228, 0.00020302128904025308,
            -0.0002589029075222507, 2.5089273607721457e-05, -0.00010985253437014219,
            -0.2503141758465685],
    44100: [-0.28459907723604855, 0.07244737666028533, -0.07937698258360983,
            0.0848775348433701, -0.08886832893562109, 0.09162147542045991,
            0.9077249116355622, 0.09162147542045991, -0.08886832893562109,
            0.0848775348433701, -0.07937698258360983, 0.07244737666028533,
            -0.28459907723604855],
}


def num_frames(length, frame_size, hop=None):
    """reference dsp/frame.py:9-10 when hop == frame_size (ceil(L/N)).
    Extension for hop < N (BASELINE shapes): frames start every `hop` samples
    and the last frame is the first one that reaches the end of the signal."""
    if hop is None or hop == frame_size:
        return int(math.ceil(float(length) / float(frame_size)))
    if length <= frame_size:
        return 1
    return 1 + int(math.ceil(float(length - frame_size) / float(hop)))


def frame_matrix(x, frame_size, hop=None):
    """reference dsp/frame.py:5-14: zero-pad the tail, split, float64 frames.
    Returns [F, N] float64 (the reference yields the rows one by one)."""
    x = np.asarray(x)
    if len(x.shape) != 1:
        raise ValueError("Only 1D numpy ndarrays are supported")
    if hop is None:
        hop = frame_size
    F = num_frames(x.shape[0], frame_size, hop)
    total = (F - 1) * hop + frame_size
    x_pad = np.concatenate((x.astype(np.float64), np.zeros(max(0, total - x.shape[0]))))
    idx = np.arange(F)[:, None] * hop + np.arange(frame_size)[None, :]
    return x_pad[idx]


def hamming_sym(n):
    """scipy.signal.hamming(n) (symmetric), reference harmonic_energy.py:42."""
    if n == 1:
        return np.ones(1)
    return 0.54 - 0.46 * np.cos(2.0 * np.pi * np.arange(n) / (n - 1))


def bark_warp_coef(fs):
    """reference dsp/wfir.py:6-10."""
    return 1.0674 * np.sqrt((2.0 / np.pi) * np.arctan(0.06583 * fs / 1000.0)) - 0.1916


def warped_remez_coefs(fs, order=12):
    """reference dsp/wfir.py:13-21; table lookup (filter *design* is a host
    constant, not hot-path arithmetic)."""
    if order != 12 or int(fs) not in REMEZ_TAPS or int(fs) != fs:
        import scipy.signal  # design-time only, for sample rates outside the table
        r = min(20000, fs / 2 - 1)
        return scipy.signal.remez(order + 1, [0, 19, 20, r, r + 1, 0.5 * fs], [0, 1, 0], fs=fs).tolist()
    return list(REMEZ_TAPS[int(fs)])


def butter2(fc, fs, btype):
    """Order-2 digital Butterworth by bilinear transform with pre-warping =
    scipy.signal.butter(2, [fc/(fs/2)], btype) (reference dsp/lowpass.py:7,
    esacf.py:133)."""
    k = math.tan(math.pi * fc / fs)
    norm = 1.0 / (1.0 + math.sqrt(2.0) * k + k * k)
    a = [1.0, 2.0 * (k * k - 1.0) * norm, (1.0 - math.sqrt(2.0) * k + k * k) * norm]
    if btype == "low":
        b = [k * k * norm, 2.0 * k * k * norm, k * k * norm]
    elif btype == "high":
        b = [norm, -2.0 * norm, norm]
    else:
        raise ValueError(btype)
    return b, a


def lfilter(b, a, x):
    """scipy.signal.lfilter (direct form II transposed, zero initial state)
    along the last axis; vectorised over leading axes."""
    x = np.asarray(x, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64) / a[0]
    a = np.asarray(a, dtype=np.float64) / a[0]
    n = max(len(a), len(b))
    b = np.concatenate([b, np.zeros(n - len(b))])
    a = np.concatenate([a, np.zeros(n - len(a))])
    z = [np.zeros(x.shape[:-1]) for _ in range(n - 1)]
    y = np.empty_like(x)
    for t in range(x.shape[-1]):
        xt = x[..., t]
        yt = z[0] + b[0] * xt if n > 1 else b[0] * xt
        for i in range(n - 2):
            z[i] = z[i + 1] + b[i + 1] * xt - a[i + 1] * yt
        if n > 1:
            z[n - 2] = b[n - 1] * xt - a[n - 1] * yt
        y[..., t] = yt
    return y


def wfir(x, fs, order=12):
    """reference dsp/wfir.py:25-43: 12 cascaded first-order all-passes
    B=[-a,1], A=[1,-a]; x_hat = c0*x + sum c[i+1]*ys[i]; returns x - x_hat."""
    a = float(bark_warp_coef(fs))
    B = [-a, 1.0]
    A = [1.0, -a]
    ys = [None] * order
    ys[0] = lfilter(B, A, x)
    for i in range(1, order):
        ys[i] = lfilter(B, A, ys[i - 1])
    c = warped_remez_coefs(fs, order)
    x_hat = c[0] * np.asarray(x, dtype=np.float64)
    for i in range(order):
        x_hat = x_hat + c[i + 1] * ys[i]
    return x - x_hat


def lowpass_filter(x, fs, band):
    """reference dsp/lowpass.py:6-8."""
    b, a = butter2(band, fs, "low")
    return lfilter(b, a, x)


def highpass_filter(x, fs):
    """reference esacf.py:132-134 (1 kHz)."""
    b, a = butter2(1000.0, fs, "high")
    return lfilter(b, a, x)
