#!/usr/bin/env python3
"""ESACF per-kernel times on the three shapes the tuning looks at (configs[2] clip batch at 44.1 and 22.05 kHz, the 8192-frame
STFT signal of the Target): best wall of five resident calls + the library's own per-kernel profile.  MPX_LIB_PATH selects
the library under test (A/B of two builds in two runs)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import numpy as np, torch
import chord_detection_amd as cd
import bench, bench_esacf as B
eng = cd.Engine(0); dev = torch.device("cuda", 0)
for label, fs, mode in (("clips 44.1 kHz", 44100, "frame"), ("clips 22.05 kHz", 22050, "frame"), ("stft 4096/1024", 44100, "stft")):
    if os.environ.get("ESACF_TIME_ONLY") and os.environ["ESACF_TIME_ONLY"] != mode:
        continue
    if mode == "stft":
        x = bench.synth_signal_device(20260101, dev); frame, hop = 4096, 1024
        if os.environ.get("ESACF_TIME_FRAMES"):   # a shorter signal: that many frames
            x = x[:(int(os.environ["ESACF_TIME_FRAMES"]) - 1) * hop + frame].contiguous()
    else:
        uniq = torch.from_numpy(B.synth_clips(fs=fs)).to(dev)
        x = uniq.repeat(4096 // 64, 1).reshape(-1).contiguous(); frame = hop = int(fs * 46.4 / 1000)
    n = x.numel(); nf = eng.num_frames(n, frame, hop)
    d_frames = torch.zeros((nf, 12), dtype=torch.float64, device=dev); d_sum = torch.zeros(12, dtype=torch.float64, device=dev)
    best = 1e9
    for i in range(6):
        t0 = time.perf_counter()
        eng.esacf_dev(x.data_ptr(), n, fs, frame, hop, d_frames.data_ptr(), d_sum.data_ptr()); eng.synchronize()
        if i: best = min(best, time.perf_counter() - t0)
    eng.profile_begin()
    eng.esacf_dev(x.data_ptr(), n, fs, frame, hop, d_frames.data_ptr(), d_sum.data_ptr()); eng.synchronize()
    prof = eng.profile_end()
    print("%-16s frames %6d  best wall %.3f ms  sum %.12e  %s" % (label, nf, 1e3 * best, float(d_sum.sum()), {k: round(v[1], 3) for k, v in prof.items()}))
