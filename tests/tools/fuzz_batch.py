#!/usr/bin/env python3
"""Batch-vs-single consistency fuzz (no oracle): for random lists of clips -- empty, one sample, ragged, longer than a
frame / chunk -- every method's batch entry point must return exactly what the single-clip entry point returns for
each clip up to the order of the final sum over frames (1e-12; ESACF in deterministic mode)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["MPX_DETERMINISTIC"] = "1"
import numpy as np
import chord_detection_amd as cd

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
eng = cd.get_engine(0)
fs = 22050
bad = 0


def clip(n):
    t = np.arange(n) / fs
    x = np.zeros(n)
    for _ in range(int(rng.integers(1, 4))):
        f0 = 440.0 * 2.0 ** ((int(rng.integers(36, 84)) - 69) / 12.0)
        for h in range(1, 5):
            x += 0.6 ** h * np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 6.28))
    return (0.3 * x + 1e-3 * rng.standard_normal(n)).astype(np.float32)


for r in range(rounds):
    lens = [int(rng.choice([0, 1, 2, 100, 1022, 1023, 1024, 2047, 8191, 8192, 8193, 20000, 44100, 70001])) for _ in range(int(rng.integers(1, 9)))]
    clips = [clip(n) for n in lens]
    checks = {
        "he": (lambda cs: eng.harmonic_energy_batch(cs, fs, 8192), lambda c: eng.harmonic_energy(c, fs, 8192)),
        "he4096/1024": (lambda cs: eng.harmonic_energy_batch(cs, fs, 4096, 1024), lambda c: eng.harmonic_energy(c, fs, 4096, 1024)),
        "he1000/300": (lambda cs: eng.harmonic_energy_batch(cs, fs, 1000, 300), lambda c: eng.harmonic_energy(c, fs, 1000, 300)),
        "esacf2227": (lambda cs: eng.esacf_batch(cs, 48000, 2227), lambda c: eng.esacf(c, 48000, 2227)),
        "esacf": (lambda cs: eng.esacf_batch(cs, fs, 1023), lambda c: eng.esacf(c, fs, 1023)),
        "if0": (lambda cs: eng.iterative_f0_batch(cs, fs), lambda c: eng.iterative_f0(c, fs)),
        "prime": (lambda cs: eng.prime_multif0_batch(cs, fs), lambda c: eng.prime_multif0(c, fs)),
    }
    for name, (fb, f1) in checks.items():
        try:
            got = fb(clips)
            want = np.stack([f1(c) for c in clips])
        except Exception as e:
            bad += 1
            print("ERROR", name, lens, repr(e)[:200])
            continue
        if not np.allclose(got, want, rtol=1e-12, atol=0):   # per-clip sums are added in a different (fixed) order
            bad += 1
            print("MISMATCH", name, lens, float(np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-300))))
print("rounds %d, failures %d" % (rounds, bad))
sys.exit(1 if bad else 0)
