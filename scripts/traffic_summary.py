#!/usr/bin/env python3
"""Per-launch FETCH_SIZE / WRITE_SIZE of the headline kernel (he_wave_kernel) from the two rocprofv3 --pmc passes of scripts/traffic_probe.py
(counter_collection.csv files given on the command line): calibration factor from the hop = N launches (known bytes),
HBM bytes per headline launch, written as text and as profiles/traffic_latest.json."""
import csv, json, os, sys
vals = {"FETCH_SIZE": [], "WRITE_SIZE": []}
for path in sys.argv[1:]:
    with open(path) as fh:
        rows = [r for r in csv.DictReader(fh) if ("he_wave_kernel" in r["Kernel_Name"] or "he_kernel" in r["Kernel_Name"]) and r["Counter_Name"] in vals]
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
    for r in rows:
        vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
fetch, write = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
cal_f, head_f = fetch[:5], fetch[5:]
cal_w, head_w = write[:5], write[5:]
known_kb = 8192 * 4096 * 4 / 1024.0
# FETCH_SIZE tallies 128-byte requests at 64 B on gfx950: x2 (MI355X_MICROARCH.md, HBM section; measured 1.9946 in round 2
# with the workgroup-per-frame kernel, whose hop = N launch reads every sample exactly once).  The wave-per-frame kernel
# prefetches unconditionally -- a wave without a next frame re-reads its last one -- so its hop = N launch is no longer a
# known-bytes launch: printed as a cross-check only.
factor = 2.0
fetch_b = factor * (sum(head_f) / len(head_f)) * 1024.0
write_b = (sum(head_w) / len(head_w)) * 1024.0 if head_w else 0.0
alg = 4144 * 8192
print("cross-check (hop = N = 4096, %.0f KB of samples + up to 25 %% re-read by the last prefetch of each wave): FETCH_SIZE %s, x2 = %.0f KB" %
      (known_kb, [round(v, 1) for v in cal_f], 2.0 * sum(cal_f) / len(cal_f)))
print("cross-check WRITE_SIZE %s (rows out: 768 KB)" % [round(v, 1) for v in cal_w])
print("headline, 18 launches rotating over 9 signals: FETCH_SIZE KB min %.1f mean %.1f max %.1f; WRITE_SIZE KB mean %.1f" %
      (min(head_f), sum(head_f) / len(head_f), max(head_f), sum(head_w) / max(len(head_w), 1)))
print("=> HBM traffic per launch: fetch %.2f MB (x2) + write %.2f MB = %.2f MB = %.3f x the algorithmic %.2f MB" %
      (fetch_b / 1e6, write_b / 1e6, (fetch_b + write_b) / 1e6, (fetch_b + write_b) / alg, alg / 1e6))
out = {"kernel": "he_wave_kernel<8,4>", "bytes_per_launch": fetch_b + write_b, "frames_per_launch": 8192,
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over scripts/traffic_probe.py (18 launches rotating "
                 "over 9 signals = 302 MB > the Infinity Cache); FETCH_SIZE x2 (MI355X_MICROARCH.md: 128-byte requests tallied at 64 B on "
                 "gfx950; 1.9946 measured on known bytes in round 2); profiles/r2/traffic_he_wave.txt", "round": 2}
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(root, "gpurun_out", "traffic_latest.json"), "w") as fh:
    json.dump(out, fh)
