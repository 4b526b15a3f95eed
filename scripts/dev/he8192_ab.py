#!/usr/bin/env python3
"""Harmonic Energy at the reference's default shape (8192-sample frames, hop = frame): the wave kernel (two passes of the 4096
pipeline per frame) against the workgroup-per-frame kernel he_kernel<8192,512,double> (mpx_set_option), same process, HIP events."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import chord_detection_amd as cd
dev = torch.device("cuda", 0)
frames, N = 8196, 8192
g = torch.Generator(device="cpu"); g.manual_seed(1)
xs = [(0.3 * torch.randn(frames * N, generator=g)).to(dev) for _ in range(2)]   # 2 x 268 MB > the Infinity Cache
rows = torch.zeros((frames, 12), dtype=torch.float64, device=dev)
res = {}
for name, opt in (("wave", 0), ("pairs", 2), ("workgroup", 1), ("wave", 0), ("pairs", 2), ("workgroup", 1)):
    e = cd.Engine(0); e.set_option("he_kernel", opt)
    for _ in range(5):
        e.harmonic_energy_dev(xs[0].data_ptr(), xs[0].numel(), 22050, N, N, rows.data_ptr(), None)
    e.synchronize(); e.timer_begin()
    reps = 40
    for r in range(reps):
        e.harmonic_energy_dev(xs[r & 1].data_ptr(), xs[0].numel(), 22050, N, N, rows.data_ptr(), None)
    ms = e.timer_end() / reps
    res.setdefault(name, []).append(ms)
    out = rows.cpu().numpy().copy()
    res.setdefault(name + "_out", out)
    print("%-10s %.2f us per %d frames  %.1f %% of 8 TB/s at 4N+48 B/frame" % (name, 1e3 * ms, frames, 100 * (4 * N + 48) * frames / (ms * 1e-3) / 8e12))
    e.close()
# the same launches on a signal that stays in L2 (hop 8: 8196 frames inside 74 KB): what the arithmetic alone takes
small = xs[0][:8196 * 8 + N].contiguous()
for name, opt in (("wave", 0), ("pairs", 2), ("workgroup", 1)):
    e = cd.Engine(0); e.set_option("he_kernel", opt)
    for _ in range(5):
        e.harmonic_energy_dev(small.data_ptr(), small.numel(), 22050, N, 8, rows.data_ptr(), None)
    e.synchronize(); e.timer_begin()
    for r in range(40):
        e.harmonic_energy_dev(small.data_ptr(), small.numel(), 22050, N, 8, rows.data_ptr(), None)
    print("%-10s %.2f us per %d frames, input resident in L2 (hop 8)" % (name, 1e3 * e.timer_end() / 40, frames))
    e.close()
print("max relative difference wave vs workgroup:", float(np.max(np.abs(res["wave_out"] - res["workgroup_out"]) / np.abs(res["workgroup_out"]))))
print("one wave per frame == a pair of waves per frame, bit for bit:", bool(np.array_equal(res["wave_out"], res["pairs_out"])))
