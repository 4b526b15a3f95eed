"""Iterative-F0 over 4096 two-second clips @22.05 kHz ALONE on the GPU: the call's wall against the sum of its kernels'
profile regions -- what part of the call is not a kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import chord_detection_amd as cd
from chord_detection_amd import corpus
fs, n = 22050, int(os.environ.get("CLIPS", "4096"))
block = corpus.synth_block(n, fs, 2.0, 1024, 0, 1, synth_device="cuda:0")
clips = block[0][1]
torch.cuda.synchronize()
eng = cd.Engine(0)
for rep in range(3):
    eng.profile_begin()
    t0 = time.perf_counter()
    eng.iterative_f0_batch(clips, fs)
    wall = time.perf_counter() - t0
    prof = eng.profile_end()
    ks = {k: round(v[1], 2) for k, v in prof.items()}
    print("call %d: wall %.1f ms, kernels %s = %.1f ms, launches %s" % (rep, 1e3 * wall, ks, sum(ks.values()), {k: v[0] for k, v in prof.items()}), flush=True)
