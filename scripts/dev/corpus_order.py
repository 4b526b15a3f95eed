"""Corpus driver (4096 two-second clips @22.05 kHz, all four methods, resident in HBM): wall and per-method completion over
several runs in ONE process (clip layout and workspaces warm: the race of corpus._start_side goes the main thread's way by
itself), side threads started at once / gated on the main context's first work.  The cold case: corpus_growth_trial.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from chord_detection_amd import corpus
fs, n = 22050, 4096
dev = "cuda:0"
corpus.run_corpus(1024, (1, 2, 3, 4), fs, 2.0, 1024, 0, 1, 0, synth_device=dev)   # warm: workspaces, plans
block = corpus.synth_block(n, fs, 2.0, 1024, 0, 1, synth_device=dev)
torch.cuda.synchronize()
ref = None
for variant in (sys.argv[1:] or ["nowait", "wait", "nowait", "wait"]):
    corpus.SIDE_THREADS_WAIT = variant == "wait"
    rows = []
    for rep in range(6):
        t0 = time.perf_counter()
        lo, hi, out, spent = corpus.run_corpus(n, (1, 2, 3, 4), fs, 2.0, 1024, 0, 1, 0, synth_device=dev, resident=block)
        rows.append((time.perf_counter() - t0, spent))
        if ref is None:
            ref = out
        assert np.array_equal(out, ref, equal_nan=True)
    print("side threads %s: wall ms %s | per method (1,2,3,4) of the median run %s" % (
        variant, " ".join("%.1f" % (1e3 * r[0]) for r in rows),
        " ".join("%.1f" % (1e3 * v) for v in sorted(rows)[len(rows) // 2][1])), flush=True)
