"""The quotient and the square root of the ESACF fit kernels (csrc/mpx_lm.hpp: lm_div, lm_sqrt) restated with exact
rational arithmetic for the fused multiply-adds, run on the CPU: from ANY estimate of 1/b or 1/sqrt(x) that is good to
the 25 bits v_rcp_f64 / v_rsq_f64 deliver (tests/tools/ubench/divacc.hip measures 2.6e8 ulp), one Newton step and the
residual step land on the correctly rounded quotient / root, or on its neighbour in rare cases -- the kernel's own
results are held to IEEE bit for bit in tests/test_gpu_lm_div_sqrt.py."""
from fractions import Fraction
import math

import numpy as np


def fma(a, b, c):
    return float(Fraction(a) * Fraction(b) + Fraction(c))   # one rounding, like v_fma_f64


def lm_div(a, b, y):
    """y: the hardware's estimate of 1/b."""
    y = fma(y, fma(-b, y, 1.0), y)
    q = a * y
    return fma(fma(-b, q, a), y, q)


def lm_sqrt(x, y):
    """y: the hardware's estimate of 1/sqrt(x)."""
    g, h = x * y, 0.5 * y
    r = fma(-h, g, 0.5)
    g = fma(g, r, g)
    h = fma(h, r, h)
    return fma(fma(-g, g, x), h, g)


def ulp(v):
    return math.ulp(v)


def test_one_newton_step_and_the_residual_step_round_correctly():
    rng = np.random.default_rng(5)
    n = 4000
    a = np.ldexp(1.0 + rng.random(n), rng.integers(-60, 60, n)) * np.where(rng.random(n) < 0.5, -1.0, 1.0)
    b = np.ldexp(1.0 + rng.random(n), rng.integers(-60, 60, n)) * np.where(rng.random(n) < 0.5, -1.0, 1.0)
    off_q = off_r = 0
    for i in range(n):
        ai, bi = float(a[i]), float(b[i])
        # an estimate with a relative error of up to 2^-25 of either sign (the measured worst case is 2^-25.1)
        err = float(rng.uniform(-1.0, 1.0)) * 2.0 ** -25
        q = lm_div(ai, bi, (1.0 / bi) * (1.0 + err))
        want = float(Fraction(ai) / Fraction(bi))
        assert abs(q - want) <= ulp(want), (ai, bi, q, want)
        off_q += q != want
        x = abs(bi)
        r = lm_sqrt(x, (1.0 / math.sqrt(x)) * (1.0 + err))
        want = math.sqrt(x)                                   # correctly rounded (IEEE)
        assert abs(r - want) <= ulp(want), (x, r, want)
        off_r += r != want
    # the residual step leaves the neighbour of the rounded result only when the exact one sits next to a rounding boundary
    assert off_q <= n // 200 and off_r <= n // 200, (off_q, off_r)


def test_small_integers_and_perfect_squares_are_exact():
    for k in range(1, 200):
        for j in (1, 2, 3, 7, 10, 21):
            want = float(Fraction(k) / Fraction(j))
            assert lm_div(float(k), float(j), (1.0 / j) * (1.0 + 2.0 ** -26)) == want, (k, j)
        x = float(k * k)
        assert lm_sqrt(x, (1.0 / k) * (1.0 - 2.0 ** -26)) == float(k)
