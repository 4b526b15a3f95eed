#!/usr/bin/env python3
"""Iterative-F0 A/B: run the 600 s stream and 1024 clips once and save the per-frame chroma (argv[1] = output .npz), so that two
builds / knob settings can be compared bit for bit:  python3 scripts/dev/if0_ab.py a.npz; MPX_...=1 python3 scripts/dev/if0_ab.py
b.npz; python3 scripts/dev/if0_ab.py a.npz b.npz  (compares)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
if len(sys.argv) == 3:
    a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
    for k in a.files:
        same = np.array_equal(a[k], b[k], equal_nan=True)
        print(k, a[k].shape, "bit-identical" if same else "DIFFERENT: max abs %.3g" % float(np.nanmax(np.abs(a[k] - b[k]))))
    sys.exit(0)
import torch
import chord_detection_amd as cd
from chord_detection_amd import corpus, stream
eng = cd.Engine(0)
dev = torch.device("cuda", 0)
out = {}
x = stream.synth_stream(0, 600 * 44100, 44100, dev)
torch.cuda.synchronize(); torch.cuda.empty_cache()
for rep in range(2):
    eng.profile_begin()
    r = eng.iterative_f0(x, 44100, frame_size=8192, return_frames=True)[1]
    prof = eng.profile_end()
out["stream600"] = np.asarray(r)
print("stream600", {k: round(v[1], 2) for k, v in prof.items()})
del x
c = corpus.synth_chunk(list(range(1024)), 22050, 2.0, dev)
for rep in range(2):
    eng.profile_begin()
    r = eng.iterative_f0_batch(c, 22050)
    prof = eng.profile_end()
out["clips1024"] = np.asarray(r)
print("clips1024", {k: round(v[1], 2) for k, v in prof.items()})
# ascii note names (every pitch class visible) on a shorter stream, and a second parameter set
x = stream.synth_stream(0, 60 * 44100, 44100, dev)
out["ascii60"] = np.asarray(eng.iterative_f0(x, 44100, frame_size=8192, return_frames=True, note_names="ascii")[1])
out["m30q25"] = np.asarray(eng.iterative_f0(x, 44100, frame_size=4096, return_frames=True, M=30, Q=25, max_voices=6)[1])
np.savez(sys.argv[1], **out)
