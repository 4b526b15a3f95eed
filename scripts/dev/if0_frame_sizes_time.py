"""Iterative-F0 over 600 s @44.1 kHz at the four tuned frame sizes: per-kernel times (MPX_LIB_PATH: A/B of two builds)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import chord_detection_amd as cd
from chord_detection_amd import stream
fs = 44100
x = stream.synth_stream(0, 600 * fs, fs, "cuda:0")
torch.cuda.synchronize()
eng = cd.Engine(0)
for nf in (8192, 4096, 2048, 1024):
    rows = torch.zeros((stream.num_frames(x.numel(), nf), 12), dtype=torch.float64, device="cuda:0")
    for rep in range(3):
        eng.profile_begin()
        eng.iterative_f0_dev(x.data_ptr(), x.numel(), fs, rows.data_ptr(), None, frame_size=nf)
        eng.synchronize()
        prof = eng.profile_end()
    print("%s frame %5d: %s  checksum %.12e" % (os.path.basename(os.environ.get("MPX_LIB_PATH", "release")), nf,
          {k: round(v[1], 2) for k, v in prof.items()}, float(rows.sum().item())), flush=True)
