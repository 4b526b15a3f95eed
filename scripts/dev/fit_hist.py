#!/usr/bin/env python3
"""Statistics build only (`make -C chord-detection_amd/csrc stats`, MPX_LIB_PATH=.../libmpx_hip_stats.so): statistics of the gaussian-fit kernels on the three ESACF
shapes of esacf_time.py -- lmpar iterations per trial, lane utilisation of the lane kernel's OUTER / INNER sections, the
cooperative kernels' trips and the wave maximum of their lmpar iterations (the 3x3 algebra is replicated in every lane)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import numpy as np, torch
import chord_detection_amd as cd
from chord_detection_amd import _lib
import bench, bench_esacf as B
lib = _lib.load()
lib.mpx_dev_fit_hist.restype = C.c_int
lib.mpx_dev_fit_hist.argtypes = [C.POINTER(C.c_uint64), C.c_int]
eng = cd.Engine(0); dev = torch.device("cuda", 0)
def hist(clear=1):
    h = np.zeros(64, dtype=np.uint64)
    assert lib.mpx_dev_fit_hist(h.ctypes.data_as(C.POINTER(C.c_uint64)), clear) == 0
    return h.astype(np.int64)
for label, fs, mode in (("clips 44.1 kHz", 44100, "frame"), ("clips 22.05 kHz", 22050, "frame"), ("stft 4096/1024", 44100, "stft")):
    if mode == "stft":
        x = bench.synth_signal_device(20260101, dev); frame, hop = 4096, 1024
    else:
        uniq = torch.from_numpy(B.synth_clips(fs=fs)).to(dev)
        x = uniq.repeat(4096 // 64, 1).reshape(-1).contiguous(); frame = hop = int(fs * 46.4 / 1000)
    n = x.numel(); nf = eng.num_frames(n, frame, hop)
    d_frames = torch.zeros((nf, 12), dtype=torch.float64, device=dev); d_sum = torch.zeros(12, dtype=torch.float64, device=dev)
    eng.esacf_dev(x.data_ptr(), n, fs, frame, hop, d_frames.data_ptr(), d_sum.data_ptr()); eng.synchronize()
    hist()
    eng.esacf_dev(x.data_ptr(), n, fs, frame, hop, d_frames.data_ptr(), d_sum.data_ptr()); eng.synchronize()
    h = hist()
    st = eng.esacf_fit_stats()
    print("== %s: frames %d, fits %s" % (label, nf, st))
    tr = h[0:12]
    print("  lane kernel: lmpar iterations per trial (0..10, 11 = initial evaluation): %s" % tr.tolist())
    print("  lane kernel: trips (waves) %d; OUTER sections %d with %.1f lanes; INNER sections %d with %.1f lanes; "
          "lmpar iterations: wave maximum %.2f per INNER section, lane mean %.2f (lane utilisation of the loop %.0f %%)"
          % (h[18], h[16], h[17] / max(h[16], 1), h[12], h[13] / max(h[12], 1), h[14] / max(h[12], 1),
             h[15] / max(h[13], 1), 100.0 * h[15] / max(64 * h[14], 1)))
    print("  cooperative: lmpar iterations per trial and fit %s; wave-trials %d, wave maximum %.2f, fits per wave-trial %.2f, "
          "trials %d" % (h[20:31].tolist(), h[32], h[33] / max(h[32], 1), h[34] / max(h[32], 1), h[35]))
