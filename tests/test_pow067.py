"""CPU: the SACF kernels' |X|^0.67 (csrc/mpx_pow067.hpp, exported for the host as mpx_test_pow067) against long-double pow.
x = |X|^2, result x^(0.67/2).  The reference computes np.abs(X) ** 0.67 in float64 (esacf.py:95-101)."""
import numpy as np


def test_pow067_against_long_double():
    from chord_detection_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(11)
    K = np.longdouble(0.5) * np.longdouble(np.float64(0.67))
    x = np.concatenate([10.0 ** rng.uniform(-30, 30, 400000) * rng.uniform(1, 2, 400000),
                        rng.uniform(1e-6, 1e6, 100000), np.array([1.0, 2.0, 4.0, 0.25, 1e-200, 1e200])])
    out = np.empty_like(x)
    assert lib.mpx_test_pow067(x.ctypes.data_as(_lib._dp), x.shape[0], out.ctypes.data_as(_lib._dp)) == 0
    ref = np.power(x.astype(np.longdouble), K)
    rel = np.abs((out.astype(np.longdouble) - ref) / ref).astype(np.float64)
    assert rel.max() <= 4e-16, (rel.max(), x[rel.argmax()])           # within an ulp and a half everywhere
    lib_path = np.exp(0.335 * np.log(x))                                  # what the kernels computed until round 3
    rel_old = np.abs((lib_path.astype(np.longdouble) - ref) / ref).astype(np.float64)
    assert rel.max() < rel_old.max()                                      # (the library path: ~4e-15 at large |log x|)
    # the reference's own float64 expression, from |X| instead of |X|^2
    mag = np.sqrt(x[:1000])
    np.testing.assert_allclose(out[:1000], mag ** 0.67, rtol=2e-15)


def test_pow067_edges():
    from chord_detection_amd import _lib
    lib = _lib.load()
    x = np.array([0.0, 5e-324, 1e-300, 1e-291, np.inf, np.nan, 1e300], dtype=np.float64)
    out = np.empty_like(x)
    assert lib.mpx_test_pow067(x.ctypes.data_as(_lib._dp), x.shape[0], out.ctypes.data_as(_lib._dp)) == 0
    assert out[0] == 0.0 and out[1] == 0.0 and out[2] == 0.0 and out[3] == 0.0     # |X| < 1e-145 counts as the exact zero
    assert np.isinf(out[4]) and np.isnan(out[5]) and out[6] == 1e300               # handed through, never garbage
    assert lib.mpx_test_pow067(None, 3, None) == _lib.MPX_EINVAL
