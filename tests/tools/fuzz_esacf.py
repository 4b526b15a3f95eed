#!/usr/bin/env python3
"""Differential fuzz of the ESACF path (lane-mode fits; FUZZ_DEFAULT_MODE=1: the default mode) against the oracle: random frame sizes, sample rates,
peak parameters and signals with silences / hard clipping (plateaus in the SACF).  Frames that differ end to end are accepted only if they agree on identical
inputs (the oracle fed the GPU's ESACF row) or the oracle flags them as ill-conditioned (perturbation test, or an
accepted gaussian fit whose centre left its 21-sample window).  Not part of the test suite
(minutes of NumPy); run on the GPU box:  python tests/tools/fuzz_esacf.py [cases] [seed]"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
if os.environ.get("FUZZ_DEFAULT_MODE") != "1":   # FUZZ_DEFAULT_MODE=1: the library's default (cooperative end game)
    os.environ["MPX_DETERMINISTIC"] = "1"
import chord_detection_amd as cd
from oracle import esacf as o_esacf

# FUZZ_WIDE=1: loud noise (more than 64 peak candidates per frame) and minimum distances up to 1000 lags (more than 31 candidates
# in range of one another): both sides of the limits of peak_pick's in-lane rounds
WIDE = os.environ.get("FUZZ_WIDE") == "1"
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
eng = cd.get_engine(0)
bad = frag = frames = 0
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for case in range(cases):
        fs = int(rng.choice([8000, 16000, 22050, 32000, 44100, 48000]))
        sizes = [int(v) for v in os.environ["FUZZ_SIZES"].split(",")] if os.environ.get("FUZZ_SIZES") else \
            [64, 100, 255, 256, 511, 512, 742, 1000, 1023, 1024, 1500, 2046, 2047, 2048, 2049, 2227, 2500, 2730, 2731, 3000, 4095, 4096, 4097, 4098, 4454, 5003, 6000, 8184, 8191, 8192, 8193, 8908, 12001, 16369, 16384]
        N = int(rng.choice(sizes))   # FUZZ_SIZES=1023,2046: the prime-factor SACF engine only
        nfr = int(rng.integers(1, 5))
        n = nfr * N - int(rng.integers(0, N // 2))
        t = np.arange(n) / fs
        x = np.zeros(n)
        for _ in range(int(rng.integers(1, 5))):
            f0 = 440.0 * 2.0 ** ((int(rng.integers(30, 90)) - 69) / 12.0)
            for h in range(1, int(rng.integers(2, 9))):
                if f0 * h < fs / 2:
                    x += rng.uniform(0.3, 1.0) ** h * np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 6.28))
        x += rng.choice([0.0, 1e-3, 0.05, 0.5, 2.0] if WIDE else [0.0, 1e-3, 0.05]) * rng.standard_normal(n)
        x *= rng.uniform(0.05, 1.0) / max(np.abs(x).max(), 1e-9)
        if rng.random() < 0.3:                      # a silent stretch
            a = int(rng.integers(0, n)); x[a:a + int(rng.integers(10, N))] = 0.0
        if rng.random() < 0.3:                      # hard clipping
            c = rng.uniform(0.2, 0.8) * np.abs(x).max(); x = np.clip(x, -c, c)
        x = x.astype(np.float32)
        kw = dict(n_peaks_elim=int(rng.integers(2, 8)), peak_thresh=float(rng.choice([0.05, 0.1, 0.3, 0.6])),
                  peak_min_dist=int(rng.choice([1, 2, 5, 10, 25, 60, 300, 1000] if WIDE else [1, 2, 5, 10, 25])), enhance_mode=str(rng.choice(["librosa010", "noop"])))
        nn = str(rng.choice(["unicode", "ascii"]))
        tot, per = eng.esacf(x, fs, N, return_frames=True, note_names=nn, **kw)
        e_gpu = eng.esacf_stage("esacf", x, fs, N, **kw)
        okw = dict(peak_thresh=kw["peak_thresh"], peak_min_dist=kw["peak_min_dist"])
        want = o_esacf.esacf_frames(x, fs, frame_size=N, n_peaks_elim=kw["n_peaks_elim"], enhance_mode=kw["enhance_mode"],
                                    note_names=nn, **okw)
        for f in range(per.shape[0]):
            frames += 1
            if np.allclose(per[f], want[f], rtol=1e-5, atol=1e-12):
                continue
            same = o_esacf.frame_chroma(e_gpu[f], fs, note_names=nn, **okw)
            if (np.allclose(per[f], same, rtol=1e-5, atol=1e-12) or o_esacf.frame_fragility(e_gpu[f], fs, **okw)
                    or o_esacf.frame_has_runaway_fit(e_gpu[f], fs, **okw)):
                frag += 1
                continue
            bad += 1
            print("MISMATCH case", case, "frame", f, fs, N, kw, per[f], want[f])
print("frames %d, equal end to end %d, equal on identical input / ill-conditioned %d, mismatches %d" % (frames, frames - frag - bad, frag, bad))
sys.exit(1 if bad else 0)
