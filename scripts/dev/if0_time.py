#!/usr/bin/env python3
"""Iterative-F0 kernel times: a 600 s stream @44.1 kHz (one piece of the 1 h run) and 1024 clips x 2 s @22.05 kHz."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import chord_detection_amd as cd
from chord_detection_amd import corpus, stream
eng = cd.Engine(0)
dev = torch.device("cuda", 0)
tag = " ".join("%s=%s" % (k, v) for k, v in os.environ.items() if k.startswith("MPX_IF0"))
x = stream.synth_stream(0, 600 * 44100, 44100, dev)
torch.cuda.synchronize(); torch.cuda.empty_cache()
r0 = eng.iterative_f0(x, 44100, frame_size=8192, return_frames=True)[1]
for rep in range(2):
    eng.profile_begin()
    t0 = time.perf_counter()
    r = eng.iterative_f0(x, 44100, frame_size=8192, return_frames=True)[1]
    dt = time.perf_counter() - t0
    prof = eng.profile_end()
print(tag, "stream600 wall %.1f ms" % (1e3 * dt), {k: round(v[1], 2) for k, v in prof.items()}, "same", bool(np.array_equal(r, r0)), "sum %.12g" % float(np.nansum(r)))
del x
NCL = int(os.environ.get("IF0_CLIPS", "1024"))
c = corpus.synth_chunk(list(range(NCL)), 22050, 2.0, dev)
r0 = eng.iterative_f0_batch(c, 22050)
for rep in range(2):
    eng.profile_begin()
    t0 = time.perf_counter()
    r = eng.iterative_f0_batch(c, 22050)
    dt = time.perf_counter() - t0
    prof = eng.profile_end()
print(tag, NCL, "clips wall %.1f ms" % (1e3 * dt), {k: round(v[1], 2) for k, v in prof.items()}, "same", bool(np.array_equal(r, r0)), "sum %.12g" % float(np.nansum(r)))
