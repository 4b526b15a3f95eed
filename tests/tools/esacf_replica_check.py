#!/usr/bin/env python3
"""ESACF end-game evidence at BASELINE configs[2] size (4096 clips, 176 573 frames): rows that differ between two
default runs, between the default (cooperative finish of the last runaway fits) and MPX_DETERMINISTIC=1 (one frame per SACF
workgroup, every fit finishes on its lane: bit-reproducible), and whether the reference algorithm itself is ill-conditioned on the
differing frames (oracle.esacf.frame_fragility: a 1e-12 relative perturbation of the ESACF row changes its chroma)."""
import os
import sys
import warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import numpy as np
import torch
import chord_detection_amd as cd
import bench_esacf as B
from oracle import esacf as o_esacf

eng = cd.Engine(0)
dev = torch.device("cuda", 0)
uniq = torch.from_numpy(B.synth_clips()).to(dev)
clips = 4096
x = uniq.repeat(clips // 64, 1).reshape(-1).contiguous()
n = x.numel()
nf = eng.num_frames(n, B.N, B.N)


def run():
    d_frames = torch.zeros((nf, 12), dtype=torch.float64, device=dev)
    d_sum = torch.zeros(12, dtype=torch.float64, device=dev)
    eng.esacf_dev(x.data_ptr(), n, B.FS, B.N, B.N, d_frames.data_ptr(), d_sum.data_ptr())
    eng.synchronize()
    return d_frames.cpu().numpy()


a, b = run(), run()
os.environ["MPX_DETERMINISTIC"] = "1"
c, c2 = run(), run()
del os.environ["MPX_DETERMINISTIC"]
rows_ab = np.flatnonzero((a != b).any(axis=1))
rows_ac = np.flatnonzero((~np.isclose(a, c, rtol=1e-9, atol=1e-12)).any(axis=1))
print("frames", nf)
print("default vs default, differing rows:", len(rows_ab))
print("lane mode vs lane mode, differing rows:", int((c != c2).any(axis=1).sum()))
print("default vs deterministic mode, rows differing beyond 1e-9:", len(rows_ac), " max |rel| elsewhere: %.1e" % float(np.max(np.abs(a - c)[np.isclose(a, c, rtol=1e-9, atol=1e-12)] / np.maximum(np.abs(c)[np.isclose(a, c, rtol=1e-9, atol=1e-12)], 1e-300))))
xh = x.cpu().numpy()
frag = 0
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for f in sorted(set(rows_ab.tolist() + rows_ac.tolist()))[:16]:
        y = eng.esacf_stage("esacf", xh[f * B.N:(f + 1) * B.N], B.FS, B.N)[0]
        fr = o_esacf.frame_fragility(y, B.FS) or o_esacf.frame_has_runaway_fit(y, B.FS)
        frag += bool(fr)
        print("  frame", f, "reference ill-conditioned (perturbation test or runaway fit):", bool(fr))
print("ill-conditioned among the differing frames checked:", frag)
