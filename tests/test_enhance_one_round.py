"""CPU: the truncation regime of the ESACF enhancement (esacf.py:108-129 with < 1024 lags, or the no-op mode) in ONE round.
csrc/mpx_esacf.hip `enhance_truncate` runs round r = 2 only -- clip, subtract the first round-half-even(Mh / 2) lags from
themselves, clip -- instead of the reference's rounds r = 2 .. n_peaks_elim: the rounds' prefixes are nested and clipping is
idempotent.  Restated here in NumPy and held to the oracle's round-by-round form (oracle/esacf.py: esacf_enhance) on every row
length of the regime, every round count, both modes, and rows with negative values, exact zeros, -0.0, NaN and inf -- equal
VALUES (NaN where the reference has NaN), and the BITS of the device's own round-by-round form (the kernels until round 6)."""
import numpy as np
import pytest

from oracle import esacf as o_esacf, thirdparty as tp


def enhance_truncate(row, n_peaks_elim, mode):
    """the device function, lag by lag"""
    v = np.array(row, dtype=np.float64)
    m = v.shape[0]
    cut2 = int(np.rint(0.5 * m)) if mode == "librosa010" else 0      # nearbyint: round half to even, like Python's round()
    if n_peaks_elim >= 2:
        with np.errstate(all="ignore"):
            v = np.where(v < 0.0, 0.0, v)
            idx = np.arange(m)
            v = np.where(idx < cut2, v - v, v)
            v = np.where(v < 0.0, 0.0, v)
    return v


def enhance_rounds(row, n_peaks_elim, mode):
    """the device code until round 6: every round, with the device's clip (`v < 0 ? 0 : v` keeps -0.0; np.clip makes it +0.0)"""
    v = np.array(row, dtype=np.float64)
    m = v.shape[0]
    idx = np.arange(m)
    with np.errstate(all="ignore"):
        for r in range(2, n_peaks_elim + 1):
            cut = int(np.rint(m / r)) if mode == "librosa010" else 0
            v = np.where(v < 0.0, 0.0, v)
            v = np.where(idx < cut, v - v, v)
            v = np.where(v < 0.0, 0.0, v)
    return v


def _bits(a):
    return np.asarray(a, dtype=np.float64).view(np.uint64)


@pytest.mark.parametrize("mode", ["librosa010", "noop"])
def test_one_round_equals_the_reference_rounds(mode):
    rng = np.random.default_rng(11)
    lengths = list(range(3, 40)) + [255, 256, 257, 510, 511, 512, 1021, 1022, 1023]
    for m in lengths:
        assert tp.time_stretch_is_truncation(m)          # the regime: every rate's stretched copy is a prefix of the row
        for n_peaks in (0, 1, 2, 3, 6, 9):
            row = rng.standard_normal(m)
            row[rng.random(m) < 0.2] = 0.0
            row[rng.random(m) < 0.05] = -0.0
            for special in (None, np.nan, np.inf, -np.inf):
                r = row.copy()
                if special is not None:
                    r[rng.integers(0, m, 3)] = special
                with np.errstate(all="ignore"):
                    want = o_esacf.esacf_enhance(r, n_peaks, mode)
                got = enhance_truncate(r, n_peaks, mode)
                # the same BITS as the device's round-by-round form (what the kernels ran until round 6) ...
                assert np.array_equal(_bits(got), _bits(enhance_rounds(r, n_peaks, mode))), (m, n_peaks, mode, special)
                # ... and the reference's values (np.clip turns a -0.0 outside the prefix into +0.0, the device keeps it: equal as
                # numbers, and nothing downstream looks at the sign of a zero), NaN where the reference has NaN
                assert np.array_equal(got, want, equal_nan=True), (m, n_peaks, mode, special)


def test_the_cuts_are_nested():
    """round-half-even(Mh / r) does not grow with r: what the one-round form rests on"""
    for m in range(1, 1200):
        cuts = [int(round(m / r)) for r in range(2, 12)]
        assert all(a >= b for a, b in zip(cuts, cuts[1:])), (m, cuts)
        assert cuts[0] == int(np.rint(0.5 * m))
