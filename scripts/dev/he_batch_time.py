#!/usr/bin/env python3
"""Harmonic Energy on one corpus chunk (1024 clips x 2 s @22.05 kHz, frame 8192): wall and per-kernel times of the batch entry."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import chord_detection_amd as cd
from chord_detection_amd import corpus
eng = cd.Engine(0)
dev = torch.device("cuda", 0)
c = corpus.synth_chunk(list(range(1024)), 22050, 2.0, dev)
for rep in range(4):
    eng.profile_begin()
    t0 = time.perf_counter()
    r = eng.harmonic_energy_batch(c, 22050)
    dt = time.perf_counter() - t0
    prof = eng.profile_end()
    print("HE batch 1024 clips: wall %.2f ms" % (1e3 * dt), {k: (v[0], round(v[1], 3)) for k, v in prof.items()})
for rep in range(3):
    eng.profile_begin()
    t0 = time.perf_counter()
    r = eng.esacf_batch(c, 22050, 1023)
    dt = time.perf_counter() - t0
    prof = eng.profile_end()
    print("ESACF batch 1024 clips: wall %.2f ms" % (1e3 * dt), {k: (v[0], round(v[1], 3)) for k, v in prof.items()})
for rep in range(3):
    eng.profile_begin()
    t0 = time.perf_counter()
    r = eng.prime_multif0_batch(c, 22050)
    dt = time.perf_counter() - t0
    prof = eng.profile_end()
    print("Prime batch 1024 clips: wall %.2f ms" % (1e3 * dt), {k: (v[0], round(v[1], 3)) for k, v in prof.items()})
