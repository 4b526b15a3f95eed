#!/usr/bin/env python3
"""Where does the headline Harmonic-Energy kernel spend its time?  Times the kernel (8192 frames, N=4096, hop 1024, rows out,
rotating over 9 signals) with parts switched off through MPX_HE_ABLATE (a profiling knob: results are garbage then):
1 no workgroup barriers in the frame loop, 2 no window/chroma tail, 4 no real-split, 8 no FFT passes after the first."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import chord_detection_amd as cd
import bench

dev = torch.device("cuda", 0)
eng = cd.Engine(0)
sigs = [bench.synth_signal_device(20260101 + k, dev) for k in range(9)]
n = sigs[0].numel()
rows = torch.empty((8192, 12), dtype=torch.float64, device=dev)

def run(reps):
    eng.timer_begin()
    for r in range(reps):
        eng.harmonic_energy_dev(sigs[r % 9].data_ptr(), n, 44100, 4096, 1024, rows.data_ptr(), None)
    return eng.timer_end() / reps * 1e3

masks = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4, 6, 7, 8, 14, 15]
res = {m: [] for m in masks}
os.environ["MPX_HE_ABLATE"] = "0"
run(2000)
for rnd in range(7):
    for m in masks:
        os.environ["MPX_HE_ABLATE"] = str(m)
        run(50)
        res[m].append(run(400))
for m in masks:
    print("ablate %2d: kernel us median %.2f min %.2f" % (m, statistics.median(res[m]), min(res[m])))
