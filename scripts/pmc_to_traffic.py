#!/usr/bin/env python3
"""profiles/traffic_latest.json (read by bench.py: `roofline.traffic` of every workload's dominant kernel) from the pmc.json
of scripts/pmc_collect.sh:  python3 scripts/pmc_to_traffic.py gpurun_out/<tag>/pmc.json [round]

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB -> B), medians over the launches of one launch shape inside one
workload of scripts/pmc_workloads.py; FETCH_SIZE and WRITE_SIZE come from separate rocprofv3 --pmc passes; the x2 is
MI355X_MICROARCH.md's gfx950 correction (128-byte requests tallied at 64 B), calibrated on known bytes in round 2 (1.9946)."""
import json
import os
import re
import sys

src = sys.argv[1]
rnd = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pmc = json.load(open(src))
out = {"round": rnd, "source": os.path.relpath(src),
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over scripts/pmc_workloads.py (bench.py's launch "
                 "shapes; the Harmonic-Energy launches rotate over 9 signals = 302 MB > the Infinity Cache); bytes = 2 x FETCH_SIZE + "
                 "WRITE_SIZE (MI355X_MICROARCH.md gfx950 correction, 1.9946 measured on known bytes in round 2); median per launch shape",
       "kernels": {}}
for key, rec in pmc.items():
    if "hbm_bytes_per_launch" not in rec:
        continue
    wl, rest = key.split("/", 1)
    name = rest.split(" grid=")[0]
    grid = int(re.search(r"grid=(\d+)", rest).group(1))
    e = out["kernels"].setdefault(wl, {}).setdefault(name, [])
    e.append({"grid": grid, "bytes_per_launch": rec["hbm_bytes_per_launch"], "launches_seen": rec["launches_seen"],
              "duration_us": rec.get("duration_us", {}).get("median")})
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
json.dump(out, open(os.path.join(root, "profiles", "traffic_latest.json"), "w"), indent=1, sort_keys=True)
print("wrote profiles/traffic_latest.json:", {w: sorted(k) for w, k in out["kernels"].items()})
