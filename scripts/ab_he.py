#!/usr/bin/env python3
"""Within-process A/B of the headline HE kernel: loads several builds of libmpx_hip.so (paths on the
command line), interleaves timed rounds on the same device buffer and prints median / min kernel time
for each (cdna guide rule 24: perf deltas come from within-probe interleaved rounds)."""
import ctypes as C, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from chord_detection_amd import _lib as L

def bind(path):
    lib = C.CDLL(path)
    for name, (res, args) in L.SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
    return lib

libs = [(p, bind(p)) for p in sys.argv[1:]]
dev = torch.device("cuda", 0)
sigs = [bench.synth_signal_device(20260101 + k, dev) for k in range(9)]; n = sigs[0].numel()   # 302 MB > the Infinity Cache
rows = torch.empty((8192, 12), dtype=torch.float64, device=dev)
sums = torch.zeros(12, dtype=torch.float64, device=dev)
ctxs = [lib.mpx_create(0, 0) for _, lib in libs]
p = L.HeParams(2, 2, 2)
for _, lib in libs:
    assert lib.mpx_abi_version() in (2, 3)
def run(lib, ctx, reps, full):
    ms = C.c_float(0)
    lib.mpx_timer_begin(ctx, None)
    for r in range(reps):
        rc = lib.mpx_harmonic_energy_dev(ctx, sigs[r % 9].data_ptr(), n, 44100, C.byref(p), 4096, 1024,
                                         None if full else rows.data_ptr(), sums.data_ptr() if full else None, None)
        assert rc == 0, lib.mpx_last_error(ctx)
    lib.mpx_timer_end(ctx, None, C.byref(ms))
    return ms.value / reps * 1e3
outs = []
for (path, lib), ctx in zip(libs, ctxs):
    run(lib, ctx, 20, False); run(lib, ctx, 20, True)
    lib.mpx_harmonic_energy_dev(ctx, sigs[0].data_ptr(), n, 44100, C.byref(p), 4096, 1024, rows.data_ptr(), None, None)
    lib.mpx_synchronize(ctx)
    outs.append(rows.clone())
for (path, _), o in zip(libs, outs):
    print("%-28s rows vs first library: max relative difference %.2e" % (os.path.basename(path), float(((o - outs[0]).abs() / outs[0].abs().clamp_min(1e-300)).max())))
res = {path: ([], []) for path, _ in libs}
for r in range(15):
    for (path, lib), ctx in zip(libs, ctxs):
        res[path][0].append(run(lib, ctx, 100, False))
        res[path][1].append(run(lib, ctx, 100, True))
ref = None
for path in res:
    k, s = res[path]
    print("%-28s kernel us median %.2f min %.2f | step us median %.2f min %.2f" % (os.path.basename(path), statistics.median(k), min(k), statistics.median(s), min(s)))
