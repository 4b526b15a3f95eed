#!/usr/bin/env python3
"""Per-frame Iterative-F0 chroma of a fixed set of inputs, saved to an .npz: run with two builds (MPX_LIB_PATH) and compare the
files bit for bit (scripts/dev/if0_bits.py out.npz; python scripts/dev/if0_bits.py --compare a.npz b.npz)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = 0
    for k in a.files:
        same = np.array_equal(a[k], b[k])
        bad += not same
        print("%-40s %s %s" % (k, a[k].shape, "equal" if same else "DIFFERENT (max abs %.3g)" % float(np.max(np.abs(a[k] - b[k])))))
    print("differing arrays:", bad)
    sys.exit(1 if bad else 0)
import torch
import chord_detection_amd as cd
from chord_detection_amd import stream, corpus
eng = cd.Engine(0); dev = torch.device("cuda", 0)
out = {}
rng = np.random.default_rng(7)
x = stream.synth_stream(0, 600 * 44100, 44100, dev)
out["stream600"] = eng.iterative_f0(x, 44100, return_frames=True)[1]
clips = torch.cat([corpus.synth_chunk(list(range(c, c + 256)), 22050, 2.0, dev) for c in range(0, 512, 256)])
out["clips512"] = eng.iterative_f0_batch(clips, 22050)
xs = x[:44100 * 20].cpu().numpy()
for fs, kw in ((44100, dict(frame_size=4096)), (48000, dict(frame_size=1024)), (22050, dict(frame_size=8192, max_voices=8)),
               (16000, dict(frame_size=2048, channels=40)), (44100, dict(frame_size=8192, power=0.5)), (96000, dict(frame_size=8192)),
               (8000, dict(frame_size=8192)), (44100, dict(frame_size=3000)), (44100, dict(frame_size=6000))):
    try:
        out["fs%d_%s" % (fs, "_".join("%s%s" % kv for kv in kw.items()))] = eng.iterative_f0(xs, fs, return_frames=True, **kw)[1]
    except Exception as e:
        print("skipped", fs, kw, type(e).__name__, str(e)[:80])
np.savez(sys.argv[1], **out)
print("saved", sys.argv[1], {k: v.shape for k, v in out.items()})
