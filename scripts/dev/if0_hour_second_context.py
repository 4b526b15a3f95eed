"""What the FIRST Iterative-F0 call of a context costs on the host when another context of the process has run the method
before (bench.py: the corpus driver's second context at 22.05 kHz, then the hour on the main one at 44.1 kHz).  MPX_IF0_TICKS=1
with the development library prints the phases of every call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import chord_detection_amd as cd
from chord_detection_amd import stream
a = cd.Engine(0)
x0 = stream.synth_stream(0, 4 * 22050, 22050, "cuda:0")
a.iterative_f0(x0, 22050)                      # another context, another sample rate
fs, secs = 44100, 3600
x = stream.synth_stream(0, secs * fs, fs, "cuda:0")
torch.cuda.synchronize(); torch.cuda.empty_cache()
b = cd.Engine(0)
b.set_option("if0_workspace_bytes", 12 << 30)
d_frames = torch.empty((stream.num_frames(x.numel(), 8192), 12), dtype=torch.float64, device="cuda:0")
for rep in range(3):
    t0 = time.perf_counter()
    b.iterative_f0_dev(x.data_ptr(), x.numel(), fs, d_frames.data_ptr(), None)
    b.synchronize()
    print("call %d: %.2f ms" % (rep, 1e3 * (time.perf_counter() - t0)), flush=True)
