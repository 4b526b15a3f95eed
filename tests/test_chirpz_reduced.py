"""The shortened chirp-z of `prime_kernel` and `he_blue_kernel` (csrc/mpx_prime.hip, csrc/mpx_he.hip), restated in NumPy:
when only the bins k < K of an N-point DFT are wanted, the convolution with the chirp spans chirp[-(N-1) .. K-1], so a
circular length L >= N + K - 1 is enough (not 2N - 1).  Same filter construction as the host code: chirp[m] at m < K,
chirp[m] at L - m for m = 1..N-1, forward FFT, pointwise product, inverse FFT, times conj(chirp[k])."""
import numpy as np
import pytest


def chirpz_low_bins(x, K, L):
    N = x.shape[0]
    assert L >= N + K - 1
    m = np.arange(N)
    chirp = np.exp(1j * np.pi * ((m * m) % (2 * N)) / N)     # e^{i pi m^2 / N}, exponent reduced like the host tables
    a = np.zeros(L, dtype=complex)
    a[:N] = x * np.conj(chirp)
    filt = np.zeros(L, dtype=complex)
    filt[:min(K, N)] = chirp[:min(K, N)]
    filt[L - np.arange(1, N)] = chirp[1:N]
    conv = np.fft.ifft(np.fft.fft(a) * np.fft.fft(filt))
    return conv[:K] * np.conj(chirp[:K])


@pytest.mark.parametrize("N,K", [(357, 90), (1348, 337), (819, 205), (5871, 1468), (5000, 901), (6500, 1172), (424, 106), (2, 1)])
def test_reduced_length_chirpz_equals_dft_on_the_wanted_bins(N, K):
    rng = np.random.default_rng(N)
    x = rng.standard_normal(N)
    L = 1024
    while L < N + K - 1:
        L *= 2
    assert L < 2 * N - 1 or N < 600          # (the point of it: a smaller transform than the full chirp-z needs)
    got = chirpz_low_bins(x, K, L)
    want = np.fft.fft(x)[:K]
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-9 * np.abs(want).max())


def test_one_bin_more_than_the_filter_holds_is_wrong():
    """The bound is tight: with a circular length of N + K - 2 the last wanted bin wraps onto the first filter tap."""
    rng = np.random.default_rng(1)
    N, K = 700, 325
    x = rng.standard_normal(N)
    m = np.arange(N)
    chirp = np.exp(1j * np.pi * ((m * m) % (2 * N)) / N)
    L = N + K - 2
    a = np.zeros(L, dtype=complex)
    a[:N] = x * np.conj(chirp)
    filt = np.zeros(L, dtype=complex)
    filt[:K] = chirp[:K]
    filt[L - np.arange(1, N)] += chirp[1:N]
    got = (np.fft.ifft(np.fft.fft(a) * np.fft.fft(filt))[:K]) * np.conj(chirp[:K])
    want = np.fft.fft(x)[:K]
    assert np.abs(got - want).max() > 1e-3 * np.abs(want).max()
