"""One call of the engine over the whole 1 h stream (89 GB of front-end output): wall and per-kernel times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import chord_detection_amd as cd
from chord_detection_amd import stream
eng = cd.Engine(0)
fs, secs = 44100, 3600
x = stream.synth_stream(0, secs * fs, fs, "cuda:0")
torch.cuda.synchronize(); torch.cuda.empty_cache()
for rep in range(3):
    eng.profile_begin()
    t0 = time.perf_counter()
    r = eng.iterative_f0(x, fs, frame_size=8192, return_frames=True)[1]
    dt = time.perf_counter() - t0
    prof = eng.profile_end()
    print("whole hour, one call: wall %.1f ms (%.0fx)" % (1e3 * dt, secs / dt), {k: round(v[1], 2) for k, v in prof.items()}, "sum %.6g" % float(np.nansum(r)))
d_frames = torch.empty((r.shape[0], 12), dtype=torch.float64, device="cuda:0")
for rep in range(3):
    t0 = time.perf_counter()
    eng.iterative_f0_dev(x.data_ptr(), x.numel(), fs, d_frames.data_ptr(), None, frame_size=8192)
    eng.synchronize()
    dt = time.perf_counter() - t0
    print("   _dev entry (frames stay in HBM): wall %.1f ms (%.0fx)" % (1e3 * dt, secs / dt))
