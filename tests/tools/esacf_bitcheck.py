#!/usr/bin/env python3
"""Are peakfit_kernel (a fit per lane) and coopfit_kernel (a fit per 16-lane row) the same arithmetic, bit for bit?
Runs ESACF batches with the runaway fits parked from different evaluation counts on (MPX_FIT_PARK_NFEV: the fits then
switch kernels at different points of their MINPACK iteration) and with parking off, and compares the per-frame chroma
rows EXACTLY.  Round 3: also with the lane kernel's two sample placements forced (MPX_FIT_SAMPLES = 0: fvec in LDS, samples re-read from
the row; 1: samples in LDS, fvec recomputed), which must be the same bits too.  The switches are development knobs: run
with MPX_LIB_PATH=chord-detection_amd/libmpx_hip_dev.so (`make dev`); the release library ignores them (the script says so).
Sizes: the BASELINE configs[2] batch (4096 clips @44.1 kHz, 2046-sample frames), the reference's own rate
(1023-sample frames @22.05 kHz) and the 8192-frame STFT signal (N=4096, hop 1024)."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import numpy as np
import torch
import chord_detection_amd as cd
import bench
import bench_esacf as B

eng = cd.Engine(0)
dev = torch.device("cuda", 0)


def run(x, fs, frame, hop, env):
    for k in ("MPX_FIT_PARK_NFEV", "MPX_FIT_NOPARK", "MPX_FIT_PARK_LIVE", "MPX_FIT_PARK_CAP", "MPX_DETERMINISTIC", "MPX_FIT_SAMPLES",
              "MPX_FIT_EARLY_NFEV", "MPX_FIT_EARLY_CAP"):
        os.environ.pop(k, None)
    os.environ.update(env)
    n = x.numel()
    nf = eng.num_frames(n, frame, hop)
    d_frames = torch.zeros((nf, 12), dtype=torch.float64, device=dev)
    d_sum = torch.zeros(12, dtype=torch.float64, device=dev)
    eng.esacf_dev(x.data_ptr(), n, fs, frame, hop, d_frames.data_ptr(), d_sum.data_ptr())
    eng.synchronize()
    t0 = time.perf_counter()
    eng.esacf_dev(x.data_ptr(), n, fs, frame, hop, d_frames.data_ptr(), d_sum.data_ptr())
    eng.synchronize()
    return d_frames.cpu().numpy(), 1e3 * (time.perf_counter() - t0)


from chord_detection_amd import _lib
print("library has development knobs:", bool(_lib.load().mpx_dev_knobs()), "(without them every row below is the default mode)")
bad = 0
for label, fs, hop_mode in (("clips 44.1 kHz", 44100, "frame"), ("clips 22.05 kHz", 22050, "frame"), ("stft 4096/1024", 44100, "stft")):
    if hop_mode == "stft":
        x = bench.synth_signal_device(20260101, dev)
        frame, hop = 4096, 1024
    else:
        uniq = torch.from_numpy(B.synth_clips(fs=fs)).to(dev)
        x = uniq.repeat(4096 // 64, 1).reshape(-1).contiguous()
        frame = hop = int(fs * 46.4 / 1000)
    ref, ms = run(x, fs, frame, hop, {"MPX_FIT_NOPARK": "1"})
    print("%-16s frames %6d  lane mode only: %.2f ms" % (label, ref.shape[0], ms))
    for env in ({}, {"MPX_FIT_PARK_NFEV": "40"}, {"MPX_FIT_PARK_NFEV": "100"}, {"MPX_FIT_PARK_NFEV": "300"},
                {"MPX_FIT_PARK_NFEV": "60", "MPX_FIT_PARK_LIVE": "64", "MPX_FIT_PARK_CAP": "100000000"},
                {"MPX_FIT_SAMPLES": "0"}, {"MPX_FIT_SAMPLES": "1"}, {"MPX_FIT_SAMPLES": "1", "MPX_FIT_NOPARK": "1"},
                {"MPX_FIT_SAMPLES": "0", "MPX_FIT_PARK_NFEV": "100"},
                # round 6: fits that look like runaways leave the lane kernel early (default for small batches only: forced
                # on for every shape here, from three points of the iteration on, with a small and the full budget; and off)
                {"MPX_FIT_EARLY_NFEV": "0"}, {"MPX_FIT_EARLY_NFEV": "8"}, {"MPX_FIT_EARLY_NFEV": "20"},
                {"MPX_FIT_EARLY_NFEV": "48", "MPX_FIT_EARLY_CAP": "700"}, {"MPX_FIT_EARLY_NFEV": "20", "MPX_FIT_PARK_NFEV": "60"}):
        got, ms = run(x, fs, frame, hop, env)
        diff = int((got != ref).any(axis=1).sum())
        bad += diff
        print("   %-70s %.2f ms   rows differing from lane mode: %d" % (env or "default", ms, diff))
print("TOTAL differing rows:", bad)
sys.exit(1 if bad else 0)
