#!/usr/bin/env python3
"""profiles/traffic_latest.json (read by bench.py: `roofline.traffic` of every workload's dominant kernel) from the pmc.json
of scripts/pmc_collect.sh:  python3 scripts/pmc_to_traffic.py gpurun_out/<tag>/pmc.json [round]

HBM-side bytes per launch = factor x FETCH_SIZE + WRITE_SIZE (KB -> B), medians over the launches of one launch shape inside
one workload of scripts/pmc_workloads.py; FETCH_SIZE and WRITE_SIZE come from separate rocprofv3 --pmc passes.

The factor: MI355X_MICROARCH.md says FETCH_SIZE reports half the bytes of WIDE coalesced reads on gfx950 (128-byte requests
tallied at 64 B) and calls every other access width uncalibrated.  Calibration on kernels whose read bytes are known
exactly (round 3, profiles/r3/pmc_after.json):
  sacf_pfa_kernel<2>  reads 16 B x 2046 x 180 224 = 5.900 GB (16 B per lane)  FETCH_SIZE 2.953 GB  -> x 1.998
  sacf_pfa_kernel<1>  reads 16 B x 1023 x 180 224 = 2.950 GB                   FETCH_SIZE 1.477 GB  -> x 1.997
  he_wave_kernel      (8 B per lane, 512 B per instruction; round 2)                                  -> x 1.995
  bandsplit_kernel    reads  4 B x 2046 x 180 224 = 1.475 GB (4 B per lane)   FETCH_SIZE 1.659 GB  -> x 0.89 (x 1 and 12 % of re-read)
  peakfit_kernel<true> reads 168 B x 2.11 M fits  = 0.354 GB (8 B per lane, scattered)  FETCH_SIZE 0.351 GB  -> x 1.01
  WRITE_SIZE: bandsplit writes 16 B x 2046 x 180 224 = 5.900 GB: WRITE_SIZE 5.906 GB; if0_frontend writes 560 B x 26.46 M
  samples = 14.82 GB: WRITE_SIZE 14.82 GB -> x 1.00
So: x 2 for the kernels that read 8 or 16 bytes per lane in full-width instructions (the transforms), x 1 for the ones that read
4-byte samples or scattered doubles (FACTOR_ONE below); the raw counters are kept in the JSON next to the result."""
import json
import os
import re
import sys

src = sys.argv[1]
rnd = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pmc = json.load(open(src))
out = {"round": rnd, "source": os.path.relpath(src),
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over scripts/pmc_workloads.py (bench.py's launch "
                 "shapes; the Harmonic-Energy launches rotate over 9 signals = 302 MB > the Infinity Cache); bytes = factor x FETCH_SIZE + "
                 "WRITE_SIZE, factor 2 for kernels reading 8-16 B per lane in full-width instructions (MI355X_MICROARCH.md gfx950 "
                 "correction; calibrated 1.995-1.998 on he_wave / sacf_pfa known bytes), 1 for 4-byte and scattered reads (calibrated "
                 "0.89-1.01 on bandsplit / peakfit known bytes): scripts/pmc_to_traffic.py; median per launch shape",
       "kernels": {}}
FACTOR_ONE = ("bandsplit_kernel", "peakfit_kernel", "coopfit_kernel", "coopfit8_kernel", "scatter_kernel", "if0_frontend_kernel", "if0_frontend2_kernel", "prime_pers_kernel", "prime_wave_kernel",
              "prime_kernel", "prime_sum_kernel", "sum_segments_kernel", "sum_chunks_kernel", "sum_all_kernel", "peakpick_kernel")
for key, rec in pmc.items():
    if "hbm_bytes_per_launch" not in rec:
        continue
    wl, rest = key.split("/", 1)
    name = rest.split(" grid=")[0]
    grid = int(re.search(r"grid=(\d+)", rest).group(1))
    e = out["kernels"].setdefault(wl, {}).setdefault(name, [])
    cs = rec["counters"]
    factor = 1.0 if name.split("<")[0] in FACTOR_ONE else 2.0
    e.append({"grid": grid, "bytes_per_launch": factor * cs["FETCH_SIZE"] * 1024.0 + cs.get("WRITE_SIZE", 0.0) * 1024.0,
              "fetch_size_bytes_raw": cs["FETCH_SIZE"] * 1024.0, "write_size_bytes": cs.get("WRITE_SIZE", 0.0) * 1024.0,
              "fetch_factor": factor, "launches_seen": rec["launches_seen"],
              "duration_us": rec.get("duration_us", {}).get("median")})
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
json.dump(out, open(os.path.join(root, "profiles", "traffic_latest.json"), "w"), indent=1, sort_keys=True)
print("wrote profiles/traffic_latest.json:", {w: sorted(k) for w, k in out["kernels"].items()})
