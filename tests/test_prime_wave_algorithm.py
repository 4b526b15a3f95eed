"""The index maps of the wave-per-item Prime-multiF0 kernel (csrc/mpx_prime.hip, prime_wave_kernel), restated in NumPy and
held against numpy.fft: two real frames per chirp-z (u = (a + i b) x window x conj(chirp)), the chirp-z as a forward and an
inverse transform of 1024 points per lane class (32 lanes x 32 registers, the layout of he_wave_kernel's transforms) --
two items per wave for the 1024-point class, one item on both classes around a radix-2 step for the 2048-point class --
the filter spectrum and the twiddles in register order, the outputs k and -k in mirrored lanes.  Runs on the CPU; the
kernel itself is covered by tests/test_gpu_prime.py."""
import numpy as np
import pytest


def _br5(p):
    return int("{:05b}".format(p)[::-1], 2)


BR = [_br5(p) for p in range(32)]


def fft1024_in_layout(z):
    """z[l, n1] = x[l + 32 n1]  ->  out[k1, p] = X[k1 + 32 br5(p)]: what transform A, the transpose and the modulated
    transform B leave in lane k1, register p (tests/test_he_wave_algorithm.py restates those steps themselves)."""
    x = np.zeros(1024, dtype=complex)
    for n1 in range(32):
        x[np.arange(32) + 32 * n1] = z[:, n1]
    X = np.fft.fft(x)
    out = np.empty((32, 32), dtype=complex)
    for p in range(32):
        out[:, p] = X[np.arange(32) + 32 * BR[p]]
    return out


def swap(v):
    return v.imag + 1j * v.real


def candidate_tables(N, L):
    """Host tables of one candidate (prime_plan / prime_wave_tables in mpx_prime.hip)."""
    half = (N // 2 + 1) // 2
    n = np.arange(N)
    win = np.hanning(N)
    chirp = np.exp(1j * np.pi * ((n * n) % (2 * N)) / N)
    filt = np.zeros(L, dtype=complex)
    filt[:max(half, 1)] = chirp[:max(half, 1)]
    m = np.arange(1, N + half - 1)                      # k - n = -1 .. -(N + half - 2)
    filt[L - m] = np.exp(1j * np.pi * ((m * m) % (2 * N)) / N)
    F = np.fft.fft(filt) / L
    wd = L // 32
    wc = np.zeros((22, wd), dtype=complex)              # [n1][lane]: window x conj(chirp) at n = lane + wd n1
    for n1 in range(22):
        idx = np.arange(wd) + wd * n1
        ok = idx < N
        wc[n1, ok] = win[idx[ok]] * np.conj(chirp[idx[ok]])
    oc = np.zeros((6, wd), dtype=complex)               # [q][lane]: conj(chirp[k])^2, k = lane + wd q
    for q in range(6):
        idx = np.arange(wd) + wd * q
        ok = idx < half
        oc[q, ok] = np.exp(-1j * np.pi * ((2 * idx[ok] * idx[ok]) % (2 * N)) / N)
    fr = np.empty((32, wd), dtype=complex)              # [p][lane]: the filter spectrum where the forward transform leaves it
    for p in range(32):
        if L == 1024:
            fr[p] = F[np.arange(32) + 32 * BR[p]]
        else:
            lane = np.arange(64)
            fr[p] = F[(lane >> 1) + 32 * BR[p] + 1024 * (lane & 1)]
    return dict(N=N, L=L, half=half, wc=wc, oc=oc, fr=fr, wsum=win.sum(), win=win)


def mirror_and_split(y, t, wd):
    """y[lane, p] = y_time[lane + wd br5(p)] -> |X_a[k]|, |X_b[k]| for k = lane + wd q, q < 6."""
    ma = np.zeros((wd, 6))
    mb = np.zeros((wd, 6))
    for lane in range(wd):
        for q in range(6):
            yp = y[lane, BR[q]]
            if lane:
                ym = y[wd - lane, BR[31 - q]]
            else:
                ym = y[0, BR[0]] if q == 0 else y[0, BR[32 - q]]
            # X[k] = conj(chirp[k]) y[k], X[-k] = conj(chirp[k]) y[-k]; X_a = (X[k] + conj X[-k]) / 2, X_b = (X[k] - conj X[-k]) / 2i:
            # times conj(chirp[k]) once more (modulus 1) they are (tt +- conj(y[-k])) / 2 with tt = conj(chirp[k])^2 y[k]
            tt = yp * t["oc"][q, lane]
            ma[lane, q], mb[lane, q] = abs(tt + np.conj(ym)), abs(tt - np.conj(ym))
    return 0.5 * ma / t["wsum"], 0.5 * mb / t["wsum"]


def wave_item_1024(a, b, t):
    N = t["N"]
    u = np.zeros(22 * 32, dtype=complex)
    u[:N] = a + 1j * b
    z = np.zeros((32, 32), dtype=complex)
    for n1 in range(22):
        z[:, n1] = u[np.arange(32) + 32 * n1] * t["wc"][n1]
    U = fft1024_in_layout(z)
    V = U * t["fr"].T                                   # [k1, p]
    w = np.empty((32, 32), dtype=complex)
    for n1 in range(32):
        w[:, n1] = swap(V[:, BR[n1]])                   # register p holds k2 = br5(p): a renaming of registers
    y = swap(fft1024_in_layout(w))                      # y[k1', p'] = y_time[k1' + 32 br5(p')]
    return mirror_and_split(y, t, 32)


def wave_item_2048(a, b, t):
    N = t["N"]
    u = np.zeros(22 * 64, dtype=complex)
    u[:N] = a + 1j * b
    z = np.zeros((64, 32), dtype=complex)
    for n1 in range(22):
        z[:, n1] = u[np.arange(64) + 64 * n1] * t["wc"][n1]
    E, O = fft1024_in_layout(z[0::2]), fft1024_in_layout(z[1::2])    # transforms of u[2m], u[2m + 1]
    k1 = np.arange(32)[:, None]
    k = k1 + 32 * np.array(BR)[None, :]
    tw2 = np.exp(-2j * np.pi * k / 2048)                # [k1, p]
    tt = tw2 * O
    Ulo, Uhi = E + tt, E - tt
    fr = t["fr"].T                                      # [lane, p]
    Vlo, Vhi = Ulo * fr[0::2], Uhi * fr[1::2]
    Ge, Go = Vlo + Vhi, (Vlo - Vhi) * np.conj(tw2)
    y = np.empty((64, 32), dtype=complex)
    for parity, G in ((0, Ge), (1, Go)):
        w = np.empty((32, 32), dtype=complex)
        for n1 in range(32):
            w[:, n1] = swap(G[:, BR[n1]])
        y[parity::2] = swap(fft1024_in_layout(w))       # y[2 k1' + parity, p'] = y_time[2 (k1' + 32 br5(p')) + parity]
    return mirror_and_split(y, t, 64)


@pytest.mark.parametrize("N,L", [(357, 1024), (500, 1024), (684, 1024), (685, 2048), (1000, 2048), (1348, 2048), (1366, 2048)])
def test_two_frames_per_chirp_z_in_the_wave_layout(N, L):
    half = (N // 2 + 1) // 2
    assert N + 2 * half - 2 <= L and 22 * (L // 32) >= N and 6 * (L // 32) >= half
    rng = np.random.default_rng(N)
    a, b = rng.standard_normal(N), rng.standard_normal(N) * 0.3
    t = candidate_tables(N, L)
    ma, mb = (wave_item_1024 if L == 1024 else wave_item_2048)(a, b, t)
    ra = np.abs(np.fft.fft(a * t["win"]))[:half] / t["wsum"]
    rb = np.abs(np.fft.fft(b * t["win"]))[:half] / t["wsum"]
    wd = L // 32
    for k in range(half):
        assert abs(ma[k % wd, k // wd] - ra[k]) < 1e-12 * ra.max()
        assert abs(mb[k % wd, k // wd] - rb[k]) < 1e-12 * ra.max()
