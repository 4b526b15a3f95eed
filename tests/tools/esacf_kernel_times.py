#!/usr/bin/env python3
"""Per-kernel HIP-event times (mpx_profile_*) of the ESACF clip batch (4096 clips, 46.4 ms frames) at 44.1 and 22.05 kHz,
with the prime-factor SACF engine (default) and the chirp-z one (MPX_SACF_BLUESTEIN=1)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import torch
import chord_detection_amd as cd
import bench_esacf as B
eng = cd.Engine(0)
dev = torch.device("cuda", 0)
for fs in (44100, 22050):
    uniq = torch.from_numpy(B.synth_clips(fs=fs)).to(dev)
    x = uniq.repeat(64, 1).reshape(-1).contiguous()
    frame = int(fs * 46.4 / 1000)
    n = x.numel(); nf = eng.num_frames(n, frame, frame)
    d_frames = torch.zeros((nf, 12), dtype=torch.float64, device=dev); d_sum = torch.zeros(12, dtype=torch.float64, device=dev)
    for env in ({}, {"MPX_SACF_BLUESTEIN": "1"}) + tuple({"MPX_SACF_ABLATE": str(m)} for m in (sys.argv[1:] and [int(v) for v in sys.argv[1:]] or [])):
        for k in ("MPX_SACF_BLUESTEIN", "MPX_SACF_ABLATE"): os.environ.pop(k, None)
        os.environ.update(env)
        for _ in range(2):
            eng.esacf_dev(x.data_ptr(), n, fs, frame, frame, d_frames.data_ptr(), d_sum.data_ptr()); eng.synchronize()
        acc = {}
        for _ in range(3):
            eng.profile_begin()
            eng.esacf_dev(x.data_ptr(), n, fs, frame, frame, d_frames.data_ptr(), d_sum.data_ptr()); eng.synchronize()
            for k, v in eng.profile_end().items(): acc[k] = acc.get(k, 0.0) + v[1] / 3
        print(fs, frame, nf, env or "pfa", {k: round(v, 3) for k, v in acc.items()}, "total", round(sum(acc.values()), 2))
