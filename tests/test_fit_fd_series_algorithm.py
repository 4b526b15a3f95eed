"""The forward-difference columns of the gaussian fits without their exponentials (csrc/mpx_esacf.hip: fd_prep / fd_row / fd_big),
restated in NumPy.  MINPACK's fdjac2 column j is (f(x + h e_j) - f(x)) / h; for the model A exp(-(x_i - c)^2 / (2 s^2 + eps)) - y_i
that is (f + y)(e^t - 1) / h with t = h (h - 2 d) ninv for the centre and t = d^2 (ninv' - ninv) for the width: held here against
the differences themselves (whose own rounding noise is eps_machine / h ~ 1e-8 of a column entry) and against the exact
derivative's first-order expansion.  Runs on the CPU; the kernels are covered by tests/test_gpu_esacf.py and esacf_bitcheck.py."""
import numpy as np

EPSMCH = np.finfo(np.float64).eps
SERIES_MAX = 0.015625


def model(p, xs):
    return p[0] * np.exp(-(xs - p[1]) ** 2 / (2.0 * p[2] ** 2 + EPSMCH))


def expm1_small(t):
    return t * (1.0 + t * (0.5 + t * (1.0 / 6.0 + t * (1.0 / 24.0 + t / 120.0))))


def fd_columns_series(p, xs, f_plus_y):
    eps = np.sqrt(EPSMCH)
    ninv = -1.0 / (2.0 * p[2] ** 2 + EPSMCH)
    h1 = eps * abs(p[1]) or eps
    h2 = eps * abs(p[2]) or eps
    h1e = (p[1] + h1) - p[1]                       # the step as taken
    s2 = p[2] + h2
    h2e = s2 - p[2]
    ninv2 = -1.0 / (2.0 * s2 * s2 + EPSMCH)
    dn = 2.0 * h2e * (p[2] + s2) * ninv * ninv2     # ninv' - ninv
    d = xs - p[1]
    t1, t2 = (h1e - 2.0 * d) * (h1e * ninv), d * d * dn
    big = max(np.abs(t1).max(), np.abs(t2).max()) > SERIES_MAX
    return f_plus_y * expm1_small(t1) / h1, f_plus_y * expm1_small(t2) / h2, big


def test_series_columns_equal_the_forward_differences_to_their_own_noise():
    rng = np.random.default_rng(5)
    checked = 0
    for _ in range(400):
        c = rng.uniform(20.0, 2000.0)
        p = np.array([rng.uniform(0.01, 5.0), c + rng.uniform(-3, 3), rng.uniform(0.6, 6.0)])
        xs = np.floor(c) - 10 + np.arange(21.0)
        y = model(p, xs) * rng.uniform(0.7, 1.3, 21) + 0.01 * rng.standard_normal(21)
        f = model(p, xs) - y
        j1, j2, big = fd_columns_series(p, xs, f + y)
        if big:
            continue
        checked += 1
        eps = np.sqrt(EPSMCH)
        for j, col in ((1, j1), (2, j2)):
            h = eps * abs(p[j])
            q = p.copy()
            q[j] = p[j] + h
            fd = ((model(q, xs) - y) - f) / h       # fdjac2
            scale = np.abs(fd).max()
            noise = 8 * EPSMCH * (np.abs(f) + np.abs(y) + p[0]).max() / h   # what the subtraction of two rounded residuals carries
            assert np.all(np.abs(col - fd) <= 1e-9 * scale + noise)
    assert checked > 300


def test_out_of_range_fits_are_flagged():
    xs = 1000.0 - 10 + np.arange(21.0)
    p = np.array([1.0, 1000.2, 0.004])              # a width far below a lag: |t| of the rows next to the centre is large
    f_plus_y = model(p, xs)
    assert fd_columns_series(p, xs, f_plus_y)[2]
    assert not fd_columns_series(np.array([1.0, 1000.2, 1.5]), xs, f_plus_y)[2]
