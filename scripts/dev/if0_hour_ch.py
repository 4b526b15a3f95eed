"""The 1 h stream with 64 / 70 channels: does the packed leftover-channel wave end the front-end launch?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import chord_detection_amd as cd
from chord_detection_amd import stream
eng = cd.Engine(0)
fs, secs = 44100, 3600
x = stream.synth_stream(0, secs * fs, fs, "cuda:0")
torch.cuda.synchronize(); torch.cuda.empty_cache()
for ch in (64, 70, 64, 70):
    eng.profile_begin()
    t0 = time.perf_counter()
    r = eng.iterative_f0(x, fs, frame_size=8192, return_frames=True, channels=ch)[1]
    dt = time.perf_counter() - t0
    prof = eng.profile_end()
    print("channels %d: wall %.1f ms" % (ch, 1e3 * dt), {k: round(v[1], 2) for k, v in prof.items()})
