"""The index maps of the wave-per-frame Harmonic-Energy kernel (csrc/mpx_he_wave.hpp), restated in NumPy and held
against numpy.fft: a 4096-point real transform as two 1024-point complex ones (one per lane parity), each 32 x 32 with the
inter-step twiddles folded into a modulated second transform, and the last two radix-2 levels evaluated only where a
window looks.  Runs on the CPU; the kernel itself is covered by tests/test_gpu_harmonic_energy.py."""
import numpy as np

N, M = 4096, 2048


def _w(n, e):
    return np.exp(-2j * np.pi * np.asarray(e) / n)


def _br5(p):
    return int("{:05b}".format(p)[::-1], 2)


def _dif32_modulated(x, theta):
    """In-register transform of the kernel: decimation in frequency of x[c] * theta^c, one constant theta^S per stage,
    outputs in bit-reversed order (hw_fft32_modulated)."""
    r = np.array(x, dtype=complex)
    s = 16
    while s >= 1:
        th = theta ** s
        for g in range(0, 32, 2 * s):
            for j in range(s):
                a, b = r[g + j], r[g + j + s]
                u = a + th * b
                d = 2 * a - u
                r[g + j] = u
                r[g + j + s] = d * _w(32, j * (16 // s))
        s //= 2
    return r


def wave_kernel_spectrum(xw):
    """ZA, ZB as the kernel's lanes hold them: lane L has the sample pairs 2(L + 64 n1), parity L & 1."""
    z = xw[0::2] + 1j * xw[1::2]                       # complex point m' = L + 64 n1
    zz = z.reshape(32, 64)                             # [n1][L]
    first = np.empty((64, 32), dtype=complex)          # A: per lane, DFT over n1 (plain DIF, bit-reversed out)
    for lane in range(64):
        first[lane] = _dif32_modulated(zz[:, lane], 1.0)
    Z = np.zeros((2, 1024), dtype=complex)
    for parity in (0, 1):
        for k1 in range(32):                           # reader lane 2 k1 + parity: row k1 of its parity class
            row = np.array([first[2 * col + parity][_br5(k1)] for col in range(32)])
            out = _dif32_modulated(row, _w(1024, k1))  # theta = W_1024^k1: the twiddles W_1024^(col k1)
            for p in range(32):
                Z[parity, k1 + 32 * _br5(p)] = out[p]
    return Z


def combine(Z, k):
    """X[k] from ZA, ZB (phase D of the kernel)."""
    kp, km = k % 1024, (1024 - k % 1024) % 1024
    w = _w(N, k)
    a, am, b, bm = Z[0, kp], np.conj(Z[0, km]), Z[1, kp], np.conj(Z[1, km])
    pa = (a + am) / 2 - 1j * w * (a - am) / 2
    pb = (b + bm) / 2 - 1j * w * (b - bm) / 2
    return pa + w * w * pb


def test_two_1024_point_transforms_and_pruned_combination_equal_rfft():
    rng = np.random.default_rng(7)
    x = rng.standard_normal(N).astype(np.float32).astype(np.float64)
    import scipy.signal.windows
    xw = x * scipy.signal.windows.hamming(N)
    Z = wave_kernel_spectrum(xw)
    np.testing.assert_allclose(Z[0], np.fft.fft(xw[0::4] + 1j * xw[1::4]), atol=1e-10)
    np.testing.assert_allclose(Z[1], np.fft.fft(xw[2::4] + 1j * xw[3::4]), atol=1e-10)
    X = np.fft.rfft(xw)
    for k in list(range(0, 40)) + list(range(95, 738)) + [1023, 1024, 1025, 2047, 2048]:
        assert abs(combine(Z, k) - X[k]) < 1e-10, k


def test_8192_sample_frame_as_two_passes_of_the_4096_pipeline():
    """HALVES = 2 of the kernel (the reference's default frame, harmonic_energy.py:14-16): x[8m + r], r < 8, as two passes,
    pass h on zA[m] = x[8m + 2h] + i x[8m + 2h + 1] and zB[m] = x[8m + 4 + 2h] + i x[8m + 5 + 2h] (sample pairs 4 pm + 2h for the
    pair index pm the lanes already use); phase D with W_8192^k, the group factor its fourth power, pass 1 folded in with its
    square -- and the window table [pass][pair] whose upper half is the mirror image of the OTHER pass."""
    import scipy.signal.windows
    n8 = 8192
    rng = np.random.default_rng(11)
    x = rng.standard_normal(n8).astype(np.float32).astype(np.float64)
    win = scipy.signal.windows.hamming(n8)
    # the window as the kernel reads it: table[h][pm < 1024] = (w[4 pm + 2h], w[4 pm + 2h + 1]); pm >= 1024 mirrored from pass 1 - h
    table = np.array([[(win[4 * pm + 2 * h], win[4 * pm + 2 * h + 1]) for pm in range(1024)] for h in range(2)])
    X = np.fft.rfft(x * win)
    Xh = []
    for h in range(2):
        sw = np.empty(4096)
        for pm in range(2048):
            w0, w1 = table[h][pm] if pm < 1024 else table[1 - h][2047 - pm][::-1]
            sw[2 * pm], sw[2 * pm + 1] = x[4 * pm + 2 * h] * w0, x[4 * pm + 2 * h + 1] * w1
        np.testing.assert_allclose(sw[0::2], (x * win)[2 * h::4], rtol=1e-14)      # the mirrored table IS the window (to the
        np.testing.assert_allclose(sw[1::2], (x * win)[2 * h + 1::4], rtol=1e-14)  # ulp by which scipy's is not symmetric)
        Z = wave_kernel_spectrum(sw)                                       # the same two 1024-point transforms per pass
        np.testing.assert_allclose(Z[0], np.fft.fft(sw[0::4] + 1j * sw[1::4]), atol=1e-10)
        Xh.append(Z)
    for k in list(range(0, 40)) + list(range(180, 1500, 7)) + [1023, 1024, 1025, 2047, 2048, 4095, 4096]:
        kp, km = k % 1024, (1024 - k % 1024) % 1024
        w = _w(n8, k)
        acc = []
        for h in range(2):
            Z = Xh[h]
            a, am, b, bm = Z[0, kp], np.conj(Z[0, km]), Z[1, kp], np.conj(Z[1, km])
            pa = (a + am) / 2 - 1j * w * (a - am) / 2
            pb = (b + bm) / 2 - 1j * w * (b - bm) / 2
            acc.append(pa + w ** 4 * pb)
        assert abs(acc[0] + w * w * acc[1] - X[k]) < 1e-9, k


def test_lds_layout_is_lane_consecutive():
    """Transpose buffer and bin-ordered copy (HW_PAIR, hw_slot): every store instruction of a wave covers 64 consecutive
    doubles, and the 64 readers of one column start in 64 different 8-byte slots of the bank sweep."""
    pair = 528
    for k1 in range(32):
        writes = sorted(8 * lane + pair * k1 for lane in range(64))
        assert writes == list(range(writes[0], writes[0] + 512, 8))
    for col in range(32):
        starts = [(pair * (lane >> 1) + 8 * (lane & 1) + 16 * col) % 512 for lane in range(32)]
        assert len(set(s // 8 for s in starts)) == 32          # 32 lanes, 32 different 8-byte slots of a 256 B x 2 sweep
    slot = lambda parity, k: 2 * (k & 1023) + parity
    for q in range(32):
        assert sorted(slot(lane & 1, (lane >> 1) + 32 * q) for lane in range(64)) == list(range(64 * q, 64 * q + 64))
