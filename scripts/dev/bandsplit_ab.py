"""x_lo / x_hi / wfir stage taps and the ESACF frames of this process's library, saved for a bit comparison between two builds;
bandsplit + whole-path timing of the Target (8192 frames of 4096 samples) and of a 4096-clip batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import chord_detection_amd as cd
from chord_detection_amd import corpus
import bench
out = sys.argv[1]
eng = cd.Engine(0)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(5)
res = {}
for fs, N, hop in ((22050, 1023, None), (44100, 2046, None), (44100, 4096, 1024), (16000, 742, None), (48000, 2227, None)):
    x = (0.3 * rng.standard_normal(N * 5 + 123)).astype(np.float32)
    for st in ("wfir", "x_lo", "x_hi", "esacf"):
        res["%d_%d_%s" % (fs, N, st)] = eng.esacf_stage(st, x, fs, N, hop)
    res["%d_%d_chroma" % (fs, N)] = eng.esacf(x, fs, N, hop, return_frames=True)[1]
np.savez(out, **res)
sig = bench.synth_signal_device(20260101, dev)
rows = torch.empty((8192, 12), dtype=torch.float64, device=dev); s12 = torch.zeros(12, dtype=torch.float64, device=dev)
for rep in range(3):
    eng.profile_begin(); t0 = time.perf_counter()
    eng.esacf_dev(sig.data_ptr(), sig.numel(), 44100, 4096, 1024, rows.data_ptr(), s12.data_ptr()); eng.synchronize()
    dt = time.perf_counter() - t0; prof = eng.profile_end()
print("target 8192 frames: wall %.2f ms" % (1e3 * dt), {k: round(v[1], 3) for k, v in prof.items()})
x = corpus.synth_chunk(list(range(64)), 44100, 2.0, dev).repeat(64, 1)[:4096].contiguous()
for rep in range(3):
    eng.profile_begin(); t0 = time.perf_counter()
    eng.esacf_batch(x, 44100, 2046)
    dt = time.perf_counter() - t0; prof = eng.profile_end()
print("clips 4096: wall %.2f ms" % (1e3 * dt), {k: round(v[1], 3) for k, v in prof.items()})
