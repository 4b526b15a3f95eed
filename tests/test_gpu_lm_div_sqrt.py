"""GPU: the quotients and square roots of the ESACF fit kernels (csrc/mpx_lm.hpp: lm_div, lm_sqrt -- v_rcp_f64 / v_rsq_f64,
one Newton step and the residual step, 7 and 10 instructions where the compiler's IEEE sequences are 12 and 17) against IEEE
division and square root (NumPy), through the C ABI's mpx_test_lm_div_sqrt.  The reference's MINPACK divides and takes
roots correctly rounded; these are held to: the same bits on every operand pair of a million drawn over 2^-300 ... 2^300,
IEEE's answers for zero, infinite and NaN operands, <= 1e-9 relative towards the ends of the exponent range (no range
scaling: documented in the header), and -- asserted, not masked -- the one corner where lm_div is NOT IEEE: a subnormal divisor
is treated as zero (+-inf where IEEE may give a finite quotient)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import chord_detection_amd as cd
    return cd.get_engine(0)


def _run(eng, a, b):
    from chord_detection_amd import _lib
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    q = np.empty_like(a)
    r = np.empty_like(a)
    rc = eng.lib.mpx_test_lm_div_sqrt(eng.ctx, a.ctypes.data_as(_lib._dp), b.ctypes.data_as(_lib._dp), a.shape[0],
                                      q.ctypes.data_as(_lib._dp), r.ctypes.data_as(_lib._dp))
    assert rc == 0, rc
    return q, r


def _operands(rng, n, emax):
    m = 1.0 + rng.random(n)
    e = rng.integers(-emax, emax + 1, n)
    s = np.where(rng.random(n) < 0.5, -1.0, 1.0)
    return s * np.ldexp(m, e)


def test_correctly_rounded_on_a_million_operand_pairs(eng):
    rng = np.random.default_rng(2024)
    n = 1 << 20
    a, b = _operands(rng, n, 300), _operands(rng, n, 300)
    q, r = _run(eng, a, np.abs(b))
    with np.errstate(all="ignore"):
        assert np.array_equal(q, a / np.abs(b))
        assert np.array_equal(r, np.sqrt(np.abs(b)))
    q, _ = _run(eng, a, b)
    assert np.array_equal(q, a / b)
    # operands a fit produces: quotients of nearly equal numbers, of small integers, reciprocals
    k = np.arange(1, 4097, dtype=np.float64)
    q, r = _run(eng, np.concatenate([k, np.ones_like(k), k + 1.0]), np.concatenate([k[::-1], k, k]))
    assert np.array_equal(q, np.concatenate([k / k[::-1], 1.0 / k, (k + 1.0) / k]))
    assert np.array_equal(r[:4096], np.sqrt(k[::-1]))


def test_zero_infinite_and_nan_operands_answer_as_ieee(eng):
    inf, nan = np.inf, np.nan
    a = np.array([1.0, -1.0, 0.0, 0.0, inf, inf, 1.0, -3.0, nan, 2.0, 0.0, -0.0, 5.0, -inf], dtype=np.float64)
    b = np.array([0.0, 0.0, 0.0, 2.0, 2.0, inf, inf, -inf, 1.0, nan, -4.0, 4.0, -0.0, -2.0], dtype=np.float64)
    q, r = _run(eng, a, b)
    with np.errstate(all="ignore"):
        wq, wr = a / b, np.sqrt(b)
    assert np.array_equal(np.isnan(q), np.isnan(wq)), (q, wq)
    ok = ~np.isnan(wq)
    assert np.array_equal(q[ok], wq[ok]) and np.array_equal(np.signbit(q[ok]), np.signbit(wq[ok])), (q, wq)
    assert np.array_equal(np.isnan(r), np.isnan(wr)), (r, wr)        # negative operands and NaN: NaN
    ok = ~np.isnan(wr)
    assert np.array_equal(r[ok], wr[ok]) and np.array_equal(np.signbit(r[ok]), np.signbit(wr[ok])), (r, wr)   # sqrt(-0) = -0


def test_towards_the_ends_of_the_exponent_range(eng):
    rng = np.random.default_rng(7)
    n = 1 << 16
    a, b = _operands(rng, n, 900), np.abs(_operands(rng, n, 900))
    q, r = _run(eng, a, b)
    with np.errstate(all="ignore"):
        wq, wr = a / b, np.sqrt(b)
    fin = np.isfinite(wq) & (np.abs(wq) > 1e-290)
    assert np.all(np.abs(q[fin] - wq[fin]) <= 1e-9 * np.abs(wq[fin]))
    assert np.all(np.abs(r - wr) <= 1e-9 * wr)
    assert eng.lib.mpx_test_lm_div_sqrt(eng.ctx, None, None, 4, None, None) == -1


def test_the_documented_corner_where_lm_div_is_not_ieee(eng):
    """csrc/mpx_lm.hpp states what the missing range scaling costs, as measured on MI355X: (1) a finite a / b that overflows
    answers +-inf like IEEE (v_div_fixup_f64 looks at the exponents); (2) a SUBNORMAL divisor is treated as zero: +-inf with
    IEEE's sign, where IEEE gives a finite quotient for a small enough numerator -- the one divergence.  If this test fails the
    arithmetic of the fit kernels changed: update the header with it.  Everything next to the corner answers as IEEE."""
    sub = 5e-310                                     # subnormal (< 2.2250738585072014e-308)
    a = np.array([1e300, -1e300, 1e-300, 1.0, -2.0, -1e-300], dtype=np.float64)
    b = np.array([1e-300, 1e-300, sub, sub, -sub, sub], dtype=np.float64)
    with np.errstate(all="ignore"):
        ieee = a / b
    assert np.array_equal(ieee[:2], [np.inf, -np.inf]) and ieee[2] == 1e-300 / sub and ieee[5] == -1e-300 / sub
    q, _ = _run(eng, a, b)
    assert np.array_equal(q[:2], ieee[:2])                                 # (1) overflow: as IEEE
    assert np.array_equal(q[3:5], ieee[3:5]) and np.all(np.isinf(ieee[3:5]))   # subnormal divisor, quotient out of range: as IEEE
    assert np.array_equal(q[[2, 5]], [np.inf, -np.inf])                    # (2) the divergence: IEEE has +-2e9 here
    # the neighbours of the corner: quotients just inside the range, the smallest NORMAL divisors, a zero divisor
    a2 = np.array([1e300, 1e-300, 3.0, 1.0, -1.0], dtype=np.float64)
    b2 = np.array([1e-7, 2.3e-308, 2.3e-308 * 4, 0.0, 0.0], dtype=np.float64)
    q2, _ = _run(eng, a2, b2)
    with np.errstate(all="ignore"):
        w2 = a2 / b2
    assert np.array_equal(np.isinf(q2), np.isinf(w2)) and np.array_equal(np.signbit(q2), np.signbit(w2)), (q2, w2)
    fin = np.isfinite(w2)
    assert np.all(np.abs(q2[fin] - w2[fin]) <= 1e-9 * np.abs(w2[fin])), (q2, w2)
