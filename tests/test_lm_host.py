"""CPU: the C++ MINPACK restatement that the GPU peak-fit kernel runs (compiled for the
host too and exported as mpx_test_gaussian_fit) against the NumPy oracle, which is itself
checked against scipy's real MINPACK in test_oracle_golden.py."""
import ctypes as C
import warnings

import numpy as np

from oracle import thirdparty as tp


def test_cpp_lm_matches_oracle_lm():
    from chord_detection_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    checked = 0
    for t in range(120):
        c = rng.uniform(300, 320)
        xs = np.arange(int(c) - 10, int(c) + 11, dtype=float)
        noise = 1e-3 if t % 2 else 2e-2
        ys = rng.uniform(0.05, 2) * np.exp(-((xs - c) ** 2) / (2 * rng.uniform(2, 30) ** 2)) + noise * rng.standard_normal(21)
        cen = C.c_double(0)
        info = lib.mpx_test_gaussian_fit(xs.ctypes.data_as(_lib._dp), ys.ctypes.data_as(_lib._dp), 21, C.byref(cen))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            p, oinfo, _ = tp.lmdif(lambda q: tp.gaussian(xs, *q) - ys, [float(ys.max()), float(xs[0]), 5.0])
        assert (info in (1, 2, 3, 4)) == (oinfo in (1, 2, 3, 4))
        if info in (1, 2, 3, 4) and xs[0] - 50 < p[1] < xs[-1] + 50:
            # both stop on ftol/xtol = 1.49e-8, so they agree to about that
            assert abs(cen.value - p[1]) <= 2e-6 * abs(p[1])
            checked += 1
    assert checked > 100
    assert lib.mpx_test_gaussian_fit(None, None, 21, None) == _lib.MPX_EINVAL
