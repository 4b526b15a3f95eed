"""Oracle: 12-bin chromagram value type, string packer, key detection.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates reference
chord_detection/chromagram.py:11-126 on plain 12-vectors.
"""
import numpy as np

NOTE_NAMES = ["C", "C#", "D", "D#", "E", "F", "F#", "G", "G#", "A", "A#", "B"]

KS_MAJOR = [6.35, 2.23, 3.48, 2.33, 4.38, 4.09, 2.52, 5.19, 2.39, 3.66, 2.29, 2.88]
KS_MINOR = [6.33, 2.68, 3.52, 5.38, 2.60, 3.53, 2.54, 4.75, 3.98, 2.69, 3.34, 3.17]


def normalize(c):
    """reference chromagram.py:50-64 on a list of 12 python floats."""
    c_ = [float(v) for v in c]
    cmin = min(c_)
    if cmin != 0.0:
        c_ = [round(v / cmin, 3) for v in c_]
    cmax = max(c_)
    if cmax > 9.0:
        c_ = [v * (9.0 / cmax) for v in c_]
    return c_


def pack(c):
    """reference chromagram.py:50-58: 12 digits via int(round(v))."""
    return "".join(str(int(round(v))) for v in normalize(c))


def _zscore(x):
    x = np.asarray(x, dtype=np.float64)
    with np.errstate(all="ignore"):
        return (x - x.mean()) / x.std()


def _circulant(c):
    c = np.asarray(c)
    n = c.shape[0]
    idx = (np.arange(n)[:, None] - np.arange(n)[None, :]) % n
    return c[idx]


def detect_key(X):
    """reference chromagram.py:84-126 (Krumhansl-Schmuckler)."""
    X = np.asarray(X)
    if X.shape[0] != 12:
        raise ValueError("input must be a chroma vector i.e. a numpy ndarray of shape (12,)")
    X = _zscore(X)
    major = _circulant(_zscore(KS_MAJOR)).T.dot(X)
    minor = _circulant(_zscore(KS_MINOR)).T.dot(X)
    mw = int(np.argmax(major) + 0.5)
    nw = int(np.argmax(minor) + 0.5)
    if major[mw] > minor[nw]:
        return "{0}maj".format(NOTE_NAMES[mw])
    elif major[mw] < minor[nw]:
        return "{0}min".format(NOTE_NAMES[nw])
    else:
        if mw == nw:
            return "{0}majmin".format(NOTE_NAMES[mw])
        return "{0}maj OR {1}min".format(NOTE_NAMES[mw], NOTE_NAMES[nw])
