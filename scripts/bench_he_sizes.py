#!/usr/bin/env python3
"""Harmonic Energy throughput for every frame size / hop the kernels cover (device-resident signal, per-frame rows
out), next to the headline configuration of bench.py.  frames/s and GB/s of algorithmic input (4*hop + 48 B/frame)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import chord_detection_amd as cd

eng = cd.Engine(0)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
out = {}
for N, hop in ((1024, 256), (2048, 512), (4096, 1024), (4096, 4096), (8192, 8192), (8192, 2048), (16384, 4096), (1000, 250), (3000, 3000)):
    F = 8192
    n = (F - 1) * hop + N
    x = torch.from_numpy((0.1 * rng.standard_normal(n)).astype(np.float32)).to(dev)
    rows = torch.empty((F, 12), dtype=torch.float64, device=dev)
    for _ in range(20):
        eng.harmonic_energy_dev(x.data_ptr(), n, 44100, N, hop, rows.data_ptr(), None)
    eng.synchronize()
    reps = 300
    eng.timer_begin()
    for _ in range(reps):
        eng.harmonic_energy_dev(x.data_ptr(), n, 44100, N, hop, rows.data_ptr(), None)
    ms = eng.timer_end() / reps
    out["N=%d hop=%d" % (N, hop)] = {"us_per_launch": 1e3 * ms, "frames_per_s": F / (ms * 1e-3),
                                      "alg_GB_per_s": F * (4 * hop + 48) / (ms * 1e-3) / 1e9}
print(json.dumps(out))
