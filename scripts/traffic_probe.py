#!/usr/bin/env python3
"""HBM-traffic probe for the HE kernel, to be run under `rocprofv3 --pmc FETCH_SIZE` (and again with WRITE_SIZE).
Launches, in order:
  5 x calibration: hop = N = 4096, 8192 frames  (every sample read exactly once: 134.2 MB known)
 18 x headline:    hop = 1024, N = 4096, 8192 frames (33.6 MB unique input each), rotating over NINE signals
                   (302 MB > the 256 MiB Infinity Cache: a launch cannot find its input in a cache warmed by the last)
All with per-frame rows out (8192*96 B = 0.79 MB written)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import chord_detection_amd as cd
import bench

eng = cd.Engine(0)
dev = torch.device("cuda", 0)
F, N = 8192, 4096
rng = np.random.default_rng(1)
big = torch.from_numpy((0.1 * rng.standard_normal(F * N)).astype(np.float32)).to(dev)
sigs = [bench.synth_signal_device(20260101 + k, dev) for k in range(9)]
rows = torch.empty((F, 12), dtype=torch.float64, device=dev)
torch.cuda.synchronize()
for _ in range(5):
    eng.harmonic_energy_dev(big.data_ptr(), big.numel(), 44100, N, N, rows.data_ptr(), None)
    eng.synchronize()
for r in range(18):
    x = sigs[r % 9]
    eng.harmonic_energy_dev(x.data_ptr(), x.numel(), 44100, N, 1024, rows.data_ptr(), None)
    eng.synchronize()
print("probe done")
