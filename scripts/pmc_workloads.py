#!/usr/bin/env python3
"""The bench's workloads once each, one launch after the other, for `rocprofv3 --pmc ...` / `--kernel-trace --stats` passes
(scripts/pmc_collect.sh): every kernel of the library appears with the launch shape it has in bench.py.

    python3 scripts/pmc_workloads.py [he,esacf_clips,esacf_1023,esacf_stft,prime,if0_clips,if0_stream]

Sizes are bench.py's (CFG / BASELINE.json configs[1..4]) except the Iterative-F0 stream: 600 s instead of 3600 s (one piece
of the 1 h run; the counters are per launch)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import chord_detection_amd as cd
from chord_detection_amd import corpus, stream
import bench

want = (sys.argv[1] if len(sys.argv) > 1 else "he,he_default,esacf_clips,esacf_1023,esacf_stft,prime,if0_clips,if0_stream").split(",")
reps = int(os.environ.get("PMC_REPS", "2"))
eng = cd.Engine(0)
dev = torch.device("cuda", 0)
FS, N, HOP, F = bench.FS, bench.N_FFT, bench.HOP, bench.FRAMES
sigs = [bench.synth_signal_device(20260101 + k, dev) for k in range(9)]
n = sigs[0].numel()
rows = torch.empty((F, 12), dtype=torch.float64, device=dev)
s12 = torch.zeros(12, dtype=torch.float64, device=dev)
torch.cuda.synchronize()

def mark(name):
    """A kernel nothing else launches (an int16 fill) separates the workloads in the profiler's dispatch order;
    scripts/pmc_report.py names what follows it after the argument order of this script."""
    torch.cuda.synchronize()
    torch.full((4096,), 1, dtype=torch.int16, device=dev)
    torch.cuda.synchronize()


order = [w for w in ("he", "he_default", "esacf_stft", "esacf_clips", "esacf_1023", "prime", "if0_clips", "if0_stream") if w in want]
print("pmc_workloads order:", ",".join(order), flush=True)
if "he" in want:
    mark("he")
    for r in range(9 * reps):
        eng.harmonic_energy_dev(sigs[r % 9].data_ptr(), n, FS, N, HOP, rows.data_ptr(), None)
        eng.synchronize()
if "he_default" in want:   # the reference's default shape: 8196 frames of 8192 samples, hop = frame, two 268 MB signals in turn
    g = torch.Generator(device="cpu"); g.manual_seed(7)
    big = [(0.3 * torch.randn(8196 * 8192, generator=g)).to(dev) for _ in range(2)]
    rows8 = torch.empty((8196, 12), dtype=torch.float64, device=dev)
    mark("he_default")
    for r in range(4 * reps):
        eng.harmonic_energy_dev(big[r & 1].data_ptr(), big[0].numel(), 22050, 8192, 8192, rows8.data_ptr(), None)
        eng.synchronize()
    del big, rows8
    torch.cuda.empty_cache()
if "esacf_stft" in want:
    mark("esacf_stft")
    for r in range(1 + reps):
        eng.esacf_dev(sigs[r % 9].data_ptr(), n, FS, N, HOP, rows.data_ptr(), s12.data_ptr())
        eng.synchronize()
for key, fs in (("esacf_clips", 44100), ("esacf_1023", 22050)):
    if key in want:
        frame = int(fs * 46.4 / 1000)
        uniq = corpus.synth_chunk(list(range(64)), fs, 2.0, dev)
        x = uniq.repeat(64, 1)[:4096].contiguous()
        mark(key)
        for r in range(1 + reps):
            eng.esacf_batch(x, fs, frame)
        del x
if "prime" in want or "if0_clips" in want:
    fs = 22050
    x = corpus.synth_chunk(list(range(1024)), fs, 2.0, dev)   # one chunk of the corpus driver (bench.py: 4 of them per GPU)
    if "prime" in want:
        mark("prime")
        for r in range(1 + reps):
            eng.prime_multif0_batch(x, fs)
    if "if0_clips" in want:
        mark("if0_clips")
        for r in range(1 + reps):
            eng.iterative_f0_batch(x, fs)
    del x
if "if0_stream" in want:
    fs = 44100
    x = stream.synth_stream(0, 600 * fs, fs, dev)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    mark("if0_stream")
    for r in range(1 + reps):
        eng.iterative_f0(x, fs, frame_size=8192)
print("pmc_workloads done:", ",".join(want))
