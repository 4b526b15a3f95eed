#!/usr/bin/env python3
"""Differential fuzz of Harmonic Energy (all five FFT sizes, hops, harmonic / octave / bin parameters, ragged lengths)
and Prime-multiF0 (parameters, lengths) against the oracle.  HE per-frame rows to 1e-9 + the rounding floor of the frame's dynamic range (frame sizes: the five powers of two and
arbitrary sizes up to 4096), Prime sums to 1e-7."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import chord_detection_amd as cd
from oracle import harmonic_energy as o_he, prime_multif0 as o_pr

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
eng = cd.get_engine(0)
bad = skipped = 0


def signal(n, fs):
    t = np.arange(n) / fs
    x = np.zeros(n)
    for _ in range(int(rng.integers(1, 5))):
        f0 = 440.0 * 2.0 ** ((int(rng.integers(30, 90)) - 69) / 12.0)
        for h in range(1, 7):
            if f0 * h < fs / 2:
                x += 0.7 ** h * np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 6.28))
    x += rng.choice([0.0, 1e-3, 0.1]) * rng.standard_normal(n)
    return (rng.uniform(0.01, 1.0) * x / max(np.abs(x).max(), 1e-9)).astype(np.float32)


with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for case in range(cases):
        fs = int(rng.choice([16000, 22050, 44100, 48000]))
        N = int(rng.choice([1024, 2048, 4096, 8192, 16384, 600, 1000, 1023, 2227, 3000, 4095, 5000, 6500, 8000, 12000, 20000, 32768]))
        hop = min(N, int(rng.choice([N, N // 2, N // 4, 1000])))
        n = int(rng.choice([N - 1, N, N + 1, 3 * N + 5, 20 * hop + N]))
        kw = dict(num_harmonic=int(rng.integers(1, 4)), num_octave=int(rng.integers(1, 4)), num_bins=int(rng.integers(0, 4)))
        x = signal(n, fs)
        try:
            want = o_he.he_frames(x.astype(np.float64), fs, N, hop, **kw)
        except Exception as e:          # parameter combinations the reference itself rejects (window beyond the spectrum)
            want = e
        try:
            tot, per = eng.harmonic_energy(x, fs, N, hop, return_frames=True, **kw)
        except NotImplementedError:     # a shape the library says it does not cover (more than 64 decimated passes)
            skipped += 1
            continue
        except Exception as e:
            per = e
        if isinstance(want, Exception) or isinstance(per, Exception):
            if isinstance(want, Exception) != isinstance(per, Exception):
                bad += 1
                print("HE error mismatch", case, fs, N, hop, n, kw, repr(want)[:80], repr(per)[:80])
            continue
        # A row entry is sqrt|X| of a window maximum: a transform's rounding error in |X| scales with the FRAME's largest bin
        # (eps * peak), so an entry far below the row maximum carries eps * rowmax^2 / (2 entry) of it (seed 551, case 214: an
        # entry 1e-4 of its row maximum, 1.9e-9 from NumPy's and 1.8e-9 from a long-double DFT, the round-4 library alike)
        ok = np.allclose(per, want, rtol=1e-9, atol=1e-12)
        if not ok and np.all(np.isfinite(want)) and np.all(np.isfinite(per)):   # (empty windows: -inf entries, compared by allclose alone)
            rowmax = np.abs(want).max(axis=-1, keepdims=True)
            floor_ = 8 * np.finfo(np.float64).eps * rowmax ** 2 / np.maximum(np.abs(want), 1e-6 * rowmax + 1e-300)
            ok = bool(np.all(np.abs(per - want) <= 1e-9 * np.abs(want) + 1e-12 + floor_))
        if not ok:
            bad += 1
            print("HE MISMATCH", case, fs, N, hop, n, kw, float(np.max(np.abs(per - want) / np.maximum(np.abs(want), 1e-300))))
    for case in range(cases // 3):
        fs = int(rng.choice([22050, 22050, 44100, 48000, 96000, 120000, 192000] + ([8000, 11025, 16000, 32000] if os.environ.get("FUZZ_PRIME_LOW_RATES") else [])))   # above ~107 kHz: decimated chirp-z passes
        n = int(rng.choice([3000, 22050, 44100, 50001])) * (1 if fs <= 48000 else 2)
        kw = dict(num_harmonic=int(rng.integers(1, 3)), num_octave=int(rng.integers(1, 4 if os.environ.get("FUZZ_PRIME_LOW_RATES") else 3)),
                  harmonic_multiples_elim=int(rng.integers(1, 7)), harmonic_elim_runs=int(rng.integers(1, 4)))
        x = signal(n, fs)
        got = eng.prime_multif0(x, fs, **kw)
        want = o_pr.prime_compute(x.astype(np.float64), fs, **kw)
        # atol: after the elimination runs the argmax walks the noise floor, where which bin wins (and so which pitch
        # class receives that ~1e-7 magnitude) is decided by rounding; everything above that level agrees to 1e-7
        if not np.allclose(got, want, rtol=1e-7, atol=1e-6):
            bad += 1
            print("PRIME MISMATCH", case, n, kw, got, want)
print("cases %d + %d, mismatches %d, unsupported shapes skipped %d" % (cases, cases // 3, bad, skipped))
sys.exit(1 if bad else 0)
