#!/usr/bin/env python3
"""Prime-multiF0 batch timing (4096 clips x 2 s @22.05 kHz, device resident) with the library's per-kernel profile."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import chord_detection_amd as cd
from chord_detection_amd import corpus
eng = cd.Engine(0)
dev = torch.device("cuda", 0)
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fs = int(sys.argv[2]) if len(sys.argv) > 2 else 22050
x = torch.cat([corpus.synth_chunk(list(range(c, min(c + 1024, clips))), fs, 2.0, dev) for c in range(0, clips, 1024)])
torch.cuda.synchronize()
r0 = eng.prime_multif0_batch(x, fs)
for rep in range(3):
    eng.profile_begin()
    t0 = time.perf_counter()
    r = eng.prime_multif0_batch(x, fs)
    dt = time.perf_counter() - t0
    prof = eng.profile_end()
    print("clips %d fs %d wall %.1f ms  " % (clips, fs, 1e3 * dt), {k: round(v[1], 2) for k, v in prof.items()}, "same", bool(np.array_equal(r, r0)))
from oracle import prime_multif0 as o
import warnings
warnings.simplefilter("ignore")
for i in (0, 5, clips - 1):
    want = o.prime_compute(x[i].cpu().numpy(), fs)
    print("oracle clip", i, "max rel", float(np.max(np.abs(r[i] - want) / (np.abs(want) + 1e-30))), "ok", bool(np.allclose(r[i], want, rtol=1e-7, atol=1e-9)))
