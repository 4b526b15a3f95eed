#!/usr/bin/env python3
"""Differential fuzz of the Iterative-F0 path against the oracle: random frame sizes (1024/2048/4096/8192 and any size 16..16384 by chirp-z), channel counts
(incl. > 64: two front-end waves, and < 64), powers, clip lengths around chunk / frame boundaries; summary spectra to
1e-9, per-frame chroma to 1e-5.  Not part of the suite (the oracle's 70-channel filterbank is seconds per clip)."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import chord_detection_amd as cd
from oracle import iterative_f0 as o

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
eng = cd.get_engine(0)
bad = skipped = 0
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for case in range(cases):
        fs = 22050
        NF = int(rng.choice([1024, 2048, 4096, 8192, int(rng.integers(16, 4096)), int(rng.integers(16, 4096)), int(rng.integers(4097, 8192)),
                             int(rng.integers(8193, 16385))]))   # tuned powers of two and chirp-z sizes (round 6: up to 16384)
        ch = int(rng.choice([5, 31, 64, 65, 70]))
        power = float(rng.choice([1.0, 0.5, 2.0]))
        n = int(rng.choice([NF // 2, NF, NF + 1, 2 * NF + 17, 16384 + 100, 3 * NF - 1, 40000, 70001, 140000, 300000]))
        t = np.arange(n) / fs
        x = np.zeros(n)
        for _ in range(int(rng.integers(1, 4))):
            f0 = 440.0 * 2.0 ** ((int(rng.integers(36, 84)) - 69) / 12.0)
            for h in range(1, 6):
                x += 0.6 ** h * np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 6.28))
        x = (0.3 * x / max(np.abs(x).max(), 1e-9) + 1e-3 * rng.standard_normal(n)).astype(np.float32)
        kw = dict(frame_size=NF, power=power, channels=ch)
        try:
            wper, wut = o.iterative_f0_frames(x, fs, **kw)
        except (IndexError, ValueError) as exc:   # the reference itself raises for this input (tiny frames: a partial's window
            skipped += 1                           # reaches bin n, periodicity.py:93-96): nothing to compare with
            print("skipped (the reference raises %s): frame %d channels %d n %d" % (type(exc).__name__, NF, ch, n))
            continue
        ut = eng.iterative_f0_spectra(x, fs, **kw)
        tot, per = eng.iterative_f0(x, fs, return_frames=True, **kw)
        ok1 = np.allclose(ut, wut, rtol=1e-9, atol=1e-9 * np.abs(wut).max())
        ok2 = np.allclose(per, wper, rtol=1e-5, atol=1e-300)
        if not (ok1 and ok2):
            bad += 1
            print("MISMATCH", case, NF, ch, power, n, "spectra", ok1, "chroma", ok2)
print("cases %d, mismatches %d, skipped because the reference raises %d" % (cases, bad, skipped))
sys.exit(1 if bad else 0)
