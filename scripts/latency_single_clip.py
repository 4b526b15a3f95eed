#!/usr/bin/env python3
"""Latency of the reference's own use case: ONE 2 s clip @22050 Hz through each method's compute_pitches()
(host array in, Chromagram out; includes PCIe and all launches).  Median of 20 calls after warm-up."""
import json, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import chord_detection_amd as cd
from chord_detection_amd import corpus

x = corpus.synth_chunk([3], 22050, 2.0).numpy()[0]
out = {}
for num, cls in cd.METHODS.items():
    obj = cls((x, 22050))
    for _ in range(3):
        c = obj.compute_pitches()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        c = obj.compute_pitches()
        ts.append(time.perf_counter() - t0)
    out[cls.display_name()] = {"median_ms": 1e3 * statistics.median(ts), "min_ms": 1e3 * min(ts), "chroma": repr(c), "key": c.key()}
print(json.dumps(out))
