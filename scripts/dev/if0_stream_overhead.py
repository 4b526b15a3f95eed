#!/usr/bin/env python3
"""Where the bench line's 1 h Iterative-F0 pass spends its time outside the engine call: cProfile of the warm pass."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from chord_detection_amd import stream
fs, secs, nf = 44100, 3600.0, 8192
n = int(round(secs * fs))
warm = stream.engine_warmup(fs, 0, frame_size=nf)
f0, f1, s0, s1, _ = stream.shard_window(n, nf, 1, 0, warm)
x = stream.synth_stream(s0, s1, fs, torch.device("cuda", 0))
torch.cuda.synchronize(); torch.cuda.empty_cache()
run = lambda: stream.run_stream_rank(lambda a, b: x[a - s0:b - s0], n, fs, 0, 1, nf, 0)[2]
for i in range(3):
    t0 = time.perf_counter(); run(); print("pass %d: %.1f ms" % (i, 1e3 * (time.perf_counter() - t0)))
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
