#!/usr/bin/env python3
"""ESACF at frame lengths above 4096 samples (sacf_split_kernel, pv_enhance_kernel<.., 4>; sacf_huge_kernel, pv_enhance_big_kernel<8>): per-kernel times of 4096 frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import chord_detection_amd as cd
eng = cd.Engine(0)
dev = torch.device("cuda", 0)
for fs, N in ((96000, 4454), (176400, 8184), (192000, 8908), (352800, 16369), (88300, 4097), (44100, 4096), (48000, 2227)):
    F = 4096
    n = F * N
    rng = np.random.default_rng(fs)
    t = np.arange(8 * N) / float(fs)
    base = sum(0.6 ** h * np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 6)) for f0 in (146.83, 220.0, 277.18) for h in range(1, 6))
    base = (0.25 * base + 0.004 * rng.standard_normal(t.shape[0])).astype(np.float32)
    x = torch.from_numpy(np.tile(base, F // 8)).to(dev)
    d_frames = torch.zeros((F, 12), dtype=torch.float64, device=dev); d_sum = torch.zeros(12, dtype=torch.float64, device=dev)
    for _ in range(2):
        eng.esacf_dev(x.data_ptr(), n, fs, N, N, d_frames.data_ptr(), d_sum.data_ptr()); eng.synchronize()
    eng.profile_begin()
    t0 = time.perf_counter()
    eng.esacf_dev(x.data_ptr(), n, fs, N, N, d_frames.data_ptr(), d_sum.data_ptr()); eng.synchronize()
    dt = time.perf_counter() - t0
    prof = eng.profile_end()
    print(fs, N, "frames", F, "wall %.2f ms = %.3g frames/s" % (1e3 * dt, F / dt), {k: round(v[1], 3) for k, v in prof.items()})
