#!/usr/bin/env python3
"""GPU occupancy of a kernel trace (rocprofv3 --kernel-trace CSV): union of the kernels' busy intervals, per queue and
overall, over the last `frac` of the trace; kernel time totals."""
import csv, sys, collections, re
path = sys.argv[1]; frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = list(csv.DictReader(open(path)))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("mpx::", ""))[:40]) for r in rows]
ev.sort()
t0, t1 = ev[0][0], max(e[1] for e in ev)
cut = t1 - frac * (t1 - t0)
ev = [e for e in ev if e[0] >= cut]
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
span = max(e[1] for e in ev) - ev[0][0]
print("window %.1f ms, kernels %d, GPU busy (union) %.1f ms = %.0f %%" % (span / 1e6, len(ev), union([(e[0], e[1]) for e in ev]) / 1e6, 100.0 * union([(e[0], e[1]) for e in ev]) / span))
byq = collections.defaultdict(list)
for e in ev: byq[e[2]].append((e[0], e[1]))
for q, iv in sorted(byq.items()): print("  queue %s: busy %.1f ms (%d kernels)" % (q, union(iv) / 1e6, len(iv)))
tot = collections.Counter(); cnt = collections.Counter()
for e in ev: tot[e[3]] += e[1] - e[0]; cnt[e[3]] += 1
for k, v in tot.most_common(18): print("  %-42s %8.2f ms  x%d" % (k, v / 1e6, cnt[k]))
# idle gaps > 0.5 ms
iv = sorted((e[0], e[1]) for e in ev); ce = iv[0][1]; gaps = []
for s, e in iv[1:]:
    if s > ce + 500000: gaps.append((s - ce) / 1e6)
    ce = max(ce, e)
print("  idle gaps > 0.5 ms:", [round(g, 2) for g in gaps])
