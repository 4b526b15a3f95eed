#!/usr/bin/env python3
"""Wall of the Harmonic-Energy batch entry at the reference's default shape (1366 clips x 6 frames of 8192 @22.05 kHz, clips
resident in HBM), 200 calls: median / best.  MPX_LIB_PATH selects the library (A/B of two builds)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import chord_detection_amd as cd
eng = cd.Engine(0); dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(5)
xs = [torch.rand((1366, 6 * 8192), generator=g, device=dev, dtype=torch.float32) - 0.5 for _ in range(2)]
for x in xs: eng.harmonic_energy_batch(x, 22050)
ts = []
for i in range(200):
    t0 = time.perf_counter(); r = eng.harmonic_energy_batch(xs[i & 1], 22050); ts.append(time.perf_counter() - t0)
ts.sort()
print("lib %s: median %.1f us, best %.1f us, checksum %.9e" % (os.path.basename(os.environ.get("MPX_LIB_PATH", "release")), 1e6 * ts[100], 1e6 * ts[0], float(np.sum(r))))
