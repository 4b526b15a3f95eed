"""The index arithmetic of the Iterative-F0 period search's range maxima (csrc/mpx_if0.hip, lane_range_max), restated in NumPy and
held against numpy.max: the residual's 8-bin block maxima (e8), a sparse table over its 64-bin blocks (st[k][b] = maximum of blocks
b .. b + 2^k - 1), and a range [lo, hi] as at most 7 + 7 bins, 7 + 7 entries of e8 and two entries of st -- what ONE LANE reads
for a range of a halving step (periodicity.py:144-163).  Runs on the CPU; the kernel itself is covered by
tests/test_gpu_iterative_f0.py."""
import numpy as np
import pytest


def build_tables(ur, nl):
    e8 = np.full((nl + 7) // 8, -np.inf)
    for b in range(len(e8)):
        e8[b] = ur[8 * b:min(8 * b + 8, nl)].max()
    nb64 = nl >> 6                                     # whole 64-bin blocks
    st = np.full((8, 256), -np.inf)
    for b in range(nb64):
        st[0, b] = ur[64 * b:64 * b + 64].max()
    for k in range(1, 8):
        half = 1 << (k - 1)
        for b in range(256):
            if b + 2 * half <= nb64:
                st[k, b] = max(st[k - 1, b], st[k - 1, b + half])
    return e8, st


def lane_range_max(ur, e8, st, lo, hi):
    mm = -np.inf
    a8, z8 = (lo + 7) >> 3, (hi + 1) >> 3
    whole8 = z8 > a8
    le, rs = (8 * a8, 8 * z8) if whole8 else (hi + 1, lo + 7)
    for j in range(7):
        if lo + j < le:
            mm = max(mm, ur[lo + j])
        if rs + j <= hi:
            mm = max(mm, ur[rs + j])
    a64, z64 = (a8 + 7) >> 3, z8 >> 3
    whole64 = whole8 and z64 > a64
    le8, rs8 = (8 * a64, 8 * z64) if whole64 else (z8, a8 + 7)
    for j in range(7):
        if whole8 and a8 + j < le8:
            mm = max(mm, e8[a8 + j])
        if whole8 and rs8 + j < z8:
            mm = max(mm, e8[rs8 + j])
    if whole64:
        k = min((z64 - a64).bit_length() - 1, 7)       # (256 blocks are two windows of 128)
        mm = max(mm, st[k, a64], st[k, z64 - (1 << k)])
    return mm


@pytest.mark.parametrize("nl", [7552, 4096, 64, 1000, 16384])
def test_one_lane_range_maximum_equals_numpy(nl):
    rng = np.random.default_rng(nl)
    ur = rng.standard_normal(nl) ** 2
    ur[rng.integers(0, nl, 40)] = 0.0
    e8, st = build_tables(ur, nl)
    cases = [(0, 0), (0, nl - 1), (nl - 1, nl - 1), (7, 8), (8, 15), (63, 64), (64, 127), (1, 14), (5, 18), (56, 72), (0, 63)]
    for _ in range(4000):
        lo = int(rng.integers(0, nl))
        span = int(rng.choice([0, 1, 6, 13, 14, 15, 40, 63, 64, 65, 127, 128, 500, 4000, nl]))
        cases.append((lo, min(nl - 1, lo + int(rng.integers(0, span + 1)))))
    for lo, hi in cases:
        if hi >= nl:
            continue
        assert lane_range_max(ur, e8, st, lo, hi) == ur[lo:hi + 1].max(), (lo, hi)
