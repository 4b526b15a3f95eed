#!/usr/bin/env python3
"""Harmonic Energy launch time vs number of frames (N=4096, hop 1024): slope = per-frame cost, intercept = the
fixed cost of a launch (dispatch, table loads, first un-prefetched frame, in-kernel reduction)."""
import ctypes as C, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from chord_detection_amd import _lib as L

lib = L.load()
dev = torch.device("cuda", 0)
ctx = lib.mpx_create(0, 0)
p = L.HeParams(2, 2, 2)
rng = np.random.default_rng(1)
pts = []
for frames in (256, 1024, 2048, 4096, 8192, 16384, 32768, 65536):
    n = (frames - 1) * 1024 + 4096
    x = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).to(dev)
    rows = torch.empty((frames, 12), dtype=torch.float64, device=dev)
    def run(reps):
        ms = C.c_float(0)
        lib.mpx_timer_begin(ctx, None)
        for _ in range(reps):
            rc = lib.mpx_harmonic_energy_dev(ctx, x.data_ptr(), n, 44100, C.byref(p), 4096, 1024, rows.data_ptr(), None, None)
            assert rc == 0, lib.mpx_last_error(ctx)
        lib.mpx_timer_end(ctx, None, C.byref(ms))
        return ms.value / reps * 1e3
    run(300)
    t = statistics.median(run(200) for _ in range(9))
    pts.append((frames, t))
    print("frames %6d  %.2f us  (%.2f ns/frame)" % (frames, t, t * 1e3 / frames))
(f0, t0), (f1, t1) = pts[4], pts[-1]
slope = (t1 - t0) / (f1 - f0)
print("slope %.3f ns/frame, intercept at 8192 frames %.2f us" % (slope * 1e3, t0 - slope * f0))
