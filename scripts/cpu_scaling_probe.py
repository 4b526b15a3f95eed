#!/usr/bin/env python3
"""How many host cores does this box really give us?  Runs the headline CPU leg of bench.py with 1..P workers."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiprocessing as mp
import bench

def main():
    info = bench.host_cpu_info()
    print(json.dumps(info))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu.stat", "/proc/self/cgroup", "/proc/loadavg"):
        try:
            print(path, open(path).read().strip().replace("\n", " | ")[:400])
        except OSError as e:
            print(path, "-", e)
    bench._CPU_INPUT["he"] = bench.synth_signal(20260101, frames=2048)
    from oracle import harmonic_energy  # noqa
    ctx = mp.get_context("fork")
    p = 1
    while p <= info["physical"]:
        t0 = time.perf_counter()
        with ctx.Pool(p) as pool:
            res = pool.map(bench._cpu_worker, [("he", 2.0, w) for w in range(p)], chunksize=1)
        print(p, "workers:", round(sum(r[0] for r in res) / max(r[1] for r in res)), "frames/s  wall", round(time.perf_counter() - t0, 2),
              "slowest worker", round(max(r[1] for r in res), 2))
        p *= 2

if __name__ == "__main__":
    main()
