#!/usr/bin/env python3
"""Per-kernel medians of the counters collected by scripts/pmc_collect.sh (<dir>/pass*.csv, <dir>/kernel_trace.csv) ->
<dir>/pmc.json and <dir>/pmc.txt.  Kernels are keyed by workload (the int16-fill markers of scripts/pmc_workloads.py
separate them in dispatch order), short name and grid.  HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (both in KB; the x2 is
MI355X_MICROARCH.md's gfx950 correction for wide coalesced reads, calibrated on known bytes in round 2: 1.9946)."""
import collections
import csv
import glob
import json
import os
import re
import sys

d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("mpx::", "")


ORDER = ["he", "he_default", "esacf_stft", "esacf_clips", "esacf_1023", "prime", "if0_clips", "if0_stream"]
order_file = os.path.join(d, "order.txt")
if os.path.exists(order_file):
    ORDER = open(order_file).read().strip().split(",")


def walk(path, name_col, handler):
    """rows of one rocprofv3 CSV in dispatch order, library kernels only, tagged with the workload they belong to"""
    with open(path) as fh:
        rows = list(csv.DictReader(fh))
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
    wl, nmark, last = "setup", 0, None
    for r in rows:
        raw = r[name_col]
        if "FillFunctor<short>" in raw or "FillFunctor<int16" in raw or "FillFunctor<signed short" in raw:
            did = r.get("Dispatch_Id")
            if did != last:   # (a counter CSV repeats the dispatch once per counter)
                wl = ORDER[nmark] if nmark < len(ORDER) else "extra%d" % nmark
                nmark += 1
                last = did
            continue
        k = short(raw)
        if not k or k.startswith("at::") or "elementwise" in k or "Memset" in k or k.startswith("__amd") or "rocprim" in k or "hipcub" in k:
            continue
        handler(wl, k, r)


def on_counter(wl, k, r):
    key = "%s/%s grid=%s wg=%s" % (wl, k, r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"))
    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))


for path in sorted(glob.glob(os.path.join(d, "pass*.csv"))):
    walk(path, "Kernel_Name", on_counter)
durs = collections.defaultdict(list)


def on_trace(wl, k, r):
    # (the trace CSV spells the launch shape per dimension)
    grid = r.get("Grid_Size") or str(int(r.get("Grid_Size_X", 1)) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1))
    wg = r.get("Workgroup_Size") or str(int(r.get("Workgroup_Size_X", 1)) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1))
    key = "%s/%s grid=%s wg=%s" % (wl, k, grid, wg)
    durs[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)


tp = os.path.join(d, "kernel_trace.csv")
if os.path.exists(tp):
    walk(tp, "Kernel_Name", on_trace)
stats = {}
sp = os.path.join(d, "kernel_stats.csv")
if os.path.exists(sp):
    with open(sp) as fh:
        for r in csv.DictReader(fh):
            stats[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6}
out = {}
lines = []
for key in sorted(acc):
    cs = {c: sorted(v)[len(v) // 2] for c, v in acc[key].items()}
    n = max(len(v) for v in acc[key].values())
    rec = {"launches_seen": n, "counters": cs}
    if "FETCH_SIZE" in cs:
        rec["hbm_bytes_per_launch"] = 2.0 * cs["FETCH_SIZE"] * 1024.0 + cs.get("WRITE_SIZE", 0.0) * 1024.0
    wc = cs.get("SQ_WAVE_CYCLES")
    if wc:
        for a, b in (("valu_active_frac", "SQ_ACTIVE_INST_VALU"), ("wait_any_frac", "SQ_WAIT_ANY"), ("wait_inst_any_frac", "SQ_WAIT_INST_ANY"),
                     ("active_any_frac", "SQ_ACTIVE_INST_ANY")):
            if b in cs:
                rec[a] = cs[b] / wc
    if cs.get("SQ_LDS_IDX_ACTIVE"):
        rec["lds_bank_conflict_frac"] = cs.get("SQ_LDS_BANK_CONFLICT", 0.0) / cs["SQ_LDS_IDX_ACTIVE"]
    if cs.get("SQ_INSTS_VALU") and cs.get("SQ_THREAD_CYCLES_VALU") and cs.get("SQ_ACTIVE_INST_VALU"):
        rec["valu_lane_utilisation"] = cs["SQ_THREAD_CYCLES_VALU"] / (64.0 * cs["SQ_ACTIVE_INST_VALU"])
    if key in durs:   # un-instrumented pass (--kernel-trace only): this launch shape's own durations
        v = sorted(durs[key])
        rec["duration_us"] = {"launches": len(v), "median": v[len(v) // 2], "min": v[0], "mean": sum(v) / len(v)}
    out[key] = rec
    lines.append(key)
    for c in sorted(cs):
        lines.append("    %-26s %.5g" % (c, cs[c]))
    if "duration_us" in rec:
        lines.append("    => duration_us median %.1f (n=%d)" % (rec["duration_us"]["median"], rec["duration_us"]["launches"]))
    for c in ("hbm_bytes_per_launch", "valu_active_frac", "wait_any_frac", "wait_inst_any_frac", "lds_bank_conflict_frac", "valu_lane_utilisation"):
        if c in rec:
            lines.append("    => %-23s %.5g" % (c, rec[c]))
json.dump(out, open(os.path.join(d, "pmc.json"), "w"), indent=1, sort_keys=True)
open(os.path.join(d, "pmc.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:400]))
