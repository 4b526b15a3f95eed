#!/usr/bin/env python3
"""Prime-multiF0 over 4096 two-second clips (device-resident), per-kernel HIP-event times, with and without the
argmax / elimination rounds (harmonic_elim_runs = 0 leaves the transforms)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import chord_detection_amd as cd
from chord_detection_amd import corpus
eng = cd.Engine(0)
clips = corpus.synth_chunk(list(range(1024)), 22050, 2.0, "cuda:0").repeat(4, 1).contiguous()
torch.cuda.synchronize()
for runs in (2, 0, 2, 0):
    eng.prime_multif0_batch(clips, 22050, harmonic_elim_runs=runs)
    eng.profile_begin()
    t0 = time.perf_counter()
    eng.prime_multif0_batch(clips, 22050, harmonic_elim_runs=runs)
    wall = time.perf_counter() - t0
    print("runs", runs, "wall %.1f ms" % (1e3 * wall), {k: round(v[1], 1) for k, v in eng.profile_end().items()})
