#!/usr/bin/env python3
"""One line per workload of a bench.py --full-json record: value, wall, per-kernel times."""
import json, sys
d = json.load(open(sys.argv[1]))
print("headline", "%.4g" % d["value"], d["unit"], "ms/step %.5f" % d["ms_per_step"], "kernel_ms %.5f" % d["roofline"]["kernel_ms"], "frac %.4f" % d["roofline"]["frac"])
for k, w in d.get("workloads", {}).items():
    ms = w.get("ms_per_batch") or (1e3 * w["wall_s"] if "wall_s" in w else None)
    print(k, "%.4g" % w["value"], w["unit"], "wall %.3f ms" % ms, {a: round(b, 3) for a, b in (w.get("kernels_ms") or w.get("kernels_ms_total") or {}).items()})
