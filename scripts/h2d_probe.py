#!/usr/bin/env python3
"""PCIe-inclusive rates of the host entry points (DESIGN.md "Host <-> device"): the headline Harmonic-Energy signal
(8192 frames, 33.5 MB) handed over as pageable host memory, as pinned host memory, as device memory through the host entry
point, and resident through the _dev entry point; and a 4096-clip ESACF batch (1.44 GB) from host memory in ONE piece and in
2 / 4 / 7 pieces whose copies overlap the kernels of the piece before (bit-equal results asserted).  The piece count is a
development switch read at mpx_create: run with MPX_LIB_PATH=chord-detection_amd/libmpx_hip_dev.so (`make dev`); with the
release library every row of the piece sweep is the default (4 pieces) and says so."""
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
for label, src in (("pageable", x_host), ("pinned", x_pin), ("device_through_host_entry", x_dev)):
    s, r = timed(lambda: eng.harmonic_energy(src, FS, N, HOP))
    assert np.array_equal(r, ref), label
    out[label] = {"ms": 1e3 * s, "frames_per_s": 8192 / s, "GB_per_s": mb / 1e3 / s}
# the same signal as 16-bit PCM (what the reference's WAV files hold): half the bytes over PCIe, x / 32768 on the device
# (include/mpx.h "PCM_16 input"); the chroma must equal the float32 entry point's on the same quantised samples bit for bit
pcm = np.clip(np.round(x_host.astype(np.float64) * 32768.0), -32768, 32767).astype(np.int16)
pcm_pin = cd.pinned_empty(pcm.shape[0], dtype=np.int16)
pcm_pin[:] = pcm
ref16 = eng.harmonic_energy(pcm.astype(np.float32) / np.float32(32768.0), FS, N, HOP)
for label, src in (("pcm16_pageable", pcm), ("pcm16_pinned", pcm_pin)):
    s, r = timed(lambda: eng.harmonic_energy(cd.Pcm16(src), FS, N, HOP))
    assert np.array_equal(r, ref16), label
    out[label] = {"ms": 1e3 * s, "frames_per_s": 8192 / s, "GB_per_s": pcm.nbytes / 1e9 / s, "bytes": int(pcm.nbytes)}
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
from chord_detection_amd import _lib
dev_build = bool(_lib.load().mpx_dev_knobs())
out["library_has_dev_knobs"] = dev_build
s, r = timed(lambda: eng.esacf_batch(clips_dev, 44100, frame), 3)
out["esacf_batch_device"] = {"ms": 1e3 * s, "clips_per_s": 4096 / s}
for pieces in (1, 2, 4, 7):
    os.environ["MPX_COPY_PIECES"] = str(pieces)   # read once, at mpx_create, by the development build only
    e = cd.Engine(0)
    s, r = timed(lambda: e.esacf_batch(clips_host, 44100, frame), 3)
    assert np.array_equal(r, want), pieces
    out["esacf_batch_pageable_pieces_%d" % pieces] = {"ms": 1e3 * s, "clips_per_s": 4096 / s, "input_GB_per_s": clips_host.nbytes / 1e9 / s,
                                                       "pieces_in_effect": pieces if dev_build else 4}
    e.close()
os.environ.pop("MPX_COPY_PIECES", None)
print(json.dumps(out, indent=1))
