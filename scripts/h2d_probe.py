#!/usr/bin/env python3
"""PCIe-inclusive rates of the host entry points (DESIGN.md "Host <-> device"): the headline Harmonic-Energy signal
(8192 frames, 33.5 MB) handed over as pageable host memory (staging ring on / off), as pinned host memory, as device
memory through the host entry point, and resident through the _dev entry point; and a 4096-clip ESACF batch (1.44 GB)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import chord_detection_amd as cd
import bench

eng = cd.Engine(0)
dev = torch.device("cuda", 0)
x_dev = bench.synth_signal_device(20260101, dev)
x_host = x_dev.cpu().numpy()
x_pin = cd.pinned_empty(x_host.shape[0])
x_pin[:] = x_host
n, mb = x_host.shape[0], x_host.nbytes / 1e6
FS, N, HOP = bench.FS, bench.N_FFT, bench.HOP


def timed(fn, reps=10):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    return (time.perf_counter() - t0) / reps, r


out = {"signal_MB": mb, "pcie_floor_ms_at_63GBps": mb / 63e3 * 1e3}
ref = eng.harmonic_energy(x_host, FS, N, HOP)
for label, src, env in (("pageable_staged", x_host, {}), ("pageable_plain_hipMemcpy", x_host, {"MPX_NO_STAGING": "1"}),
                        ("pinned", x_pin, {}), ("device_through_host_entry", x_dev, {})):
    os.environ.pop("MPX_NO_STAGING", None)
    os.environ.update(env)
    s, r = timed(lambda: eng.harmonic_energy(src, FS, N, HOP))
    assert np.array_equal(r, ref), label
    out[label] = {"ms": 1e3 * s, "frames_per_s": 8192 / s, "GB_per_s": mb / 1e3 / s}
os.environ.pop("MPX_NO_STAGING", None)
d_sum = torch.zeros(12, dtype=torch.float64, device=dev)
def resident():
    eng.harmonic_energy_dev(x_dev.data_ptr(), n, FS, N, HOP, None, d_sum.data_ptr())
    eng.synchronize()
s, _ = timed(resident, 50)
out["resident_dev_entry"] = {"ms": 1e3 * s, "frames_per_s": 8192 / s}

# ESACF clip batch from the host: 4096 clips x 2 s @44.1 kHz
from chord_detection_amd import corpus
clips_dev = corpus.synth_chunk(list(range(64)), 44100, 2.0, dev).repeat(64, 1).contiguous()
clips_host = clips_dev.cpu().numpy()
frame = int(44100 * 46.4 / 1000)
want = eng.esacf_batch(clips_dev, 44100, frame)
for label, src, env in (("esacf_batch_pageable_staged", clips_host, {}), ("esacf_batch_pageable_plain", clips_host, {"MPX_NO_STAGING": "1"}),
                        ("esacf_batch_device", clips_dev, {})):
    os.environ.pop("MPX_NO_STAGING", None)
    os.environ.update(env)
    s, r = timed(lambda: eng.esacf_batch(src, 44100, frame), 3)
    assert np.array_equal(r, want), label
    out[label] = {"ms": 1e3 * s, "clips_per_s": 4096 / s, "input_GB_per_s": clips_host.nbytes / 1e9 / s}
print(json.dumps(out, indent=1))
