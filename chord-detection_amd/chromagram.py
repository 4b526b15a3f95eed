"""Host side of the path: the 12-bin Chromagram value type, the 12-digit string
packer and Krumhansl-Schmuckler key detection.

Mirrors the reference's chord_detection/chromagram.py:11-126 interface (same
names, same rounding, same error text).  Twelve numbers per clip: this is
boundary glue, not hot-path work, and stays in Python like the reference's.
"""
from collections.abc import Sequence

import numpy

_note_names = ["C", "C#", "D", "D#", "E", "F", "F#", "G", "G#", "A", "A#", "B"]
_index = {n: i for i, n in enumerate(_note_names)}


class Chromagram(Sequence):
    """12 float bins keyed C, C#, ... B (reference chromagram.py:11-48)."""

    def __init__(self, values=None):
        self.v = numpy.zeros(12, dtype=numpy.float64)
        if values is not None:
            values = numpy.asarray(values, dtype=numpy.float64)
            if values.shape != (12,):
                raise ValueError("a chromagram holds exactly 12 bins")
            self.v[:] = values
        # The reference's __setitem__ does not map the unicode sharp that its
        # __getitem__ accepts (chromagram.py:21 vs :29), so writes under "C♯"
        # land in a stray key that nothing reads.  Kept for fidelity.
        self._stray = {}

    def __getitem__(self, i):
        if type(i) == str:
            return float(self.v[_index[i.replace("♯", "#")]])
        elif type(i) == int:
            return float(self.v[i])
        raise ValueError("this shouldn't happen")

    def __setitem__(self, i, item):
        if type(i) == str:
            if i in _index:
                self.v[_index[i]] = item
            else:
                self._stray[i] = item
        elif type(i) == int:
            self.v[i] = item
        else:
            raise ValueError("this shouldn't happen")

    def __len__(self):
        return 12 + len(self._stray)

    def __repr__(self):
        return self._pack()

    def __add__(self, other):
        self.v += other.v  # in place and returns self, like chromagram.py:42-45
        return self

    def as_array(self):
        return self.v.copy()

    def key(self):
        return detect_key(self.v.copy())

    def _pack(self):
        return "".join(str(int(round(x))) for x in _normalize([float(t) for t in self.v]))


def _normalize(values):
    """chromagram.py:61-74: divide by the minimum (3 dp) unless it is 0, then
    rescale so the maximum is 9 if it exceeds 9."""
    out = list(values)
    lo = min(out)
    if lo != 0.0:
        out = [round(x / lo, 3) for x in out]
    hi = max(out)
    if hi > 9.0:
        out = [x * (9.0 / hi) for x in out]
    return out


_KS_MAJOR = [6.35, 2.23, 3.48, 2.33, 4.38, 4.09, 2.52, 5.19, 2.39, 3.66, 2.29, 2.88]
_KS_MINOR = [6.33, 2.68, 3.52, 5.38, 2.60, 3.53, 2.54, 4.75, 3.98, 2.69, 3.34, 3.17]


def _zscore(a):
    a = numpy.asarray(a, dtype=numpy.float64)
    with numpy.errstate(all="ignore"):
        return (a - a.mean()) / a.std()


def _rotation_scores(profile, X):
    # circulant(profile).T.dot(X): score[r] = sum_i profile[(i - r) % 12] * X[i]
    p = _zscore(profile)
    idx = (numpy.arange(12)[:, None] - numpy.arange(12)[None, :]) % 12  # circulant(p)[i, j] = p[(i - j) % 12]
    return p[idx].T.dot(X)


def detect_key(X):
    """Krumhansl-Schmuckler key label (reference chromagram.py:84-126)."""
    X = numpy.asarray(X)
    if X.shape[0] != 12:
        raise ValueError(
            "input must be a chroma vector i.e. a numpy ndarray of shape (12,)"
        )
    X = _zscore(X)
    major = _rotation_scores(_KS_MAJOR, X)
    minor = _rotation_scores(_KS_MINOR, X)
    major_winner = int(numpy.argmax(major) + 0.5)
    minor_winner = int(numpy.argmax(minor) + 0.5)
    if major[major_winner] > minor[minor_winner]:
        return "{0}maj".format(_note_names[major_winner])
    elif major[major_winner] < minor[minor_winner]:
        return "{0}min".format(_note_names[minor_winner])
    if major_winner == minor_winner:
        return "{0}majmin".format(_note_names[major_winner])
    return "{0}maj OR {1}min".format(_note_names[major_winner], _note_names[minor_winner])
