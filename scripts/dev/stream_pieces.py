"""First-pass and warm wall time of the 1 h Iterative-F0 stream for a given piece size (GiB of front-end output per piece):
one fresh process per size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import chord_detection_amd as cd
from chord_detection_amd import stream
gib = float(sys.argv[1]); sub = int(sys.argv[2]) if len(sys.argv) > 2 else 3
stream.PIECE_BYTES = int(gib * (1 << 30))
fs, secs = 44100, 3600
n = secs * fs
x = stream.synth_stream(0, n, fs, "cuda:0")
torch.cuda.synchronize(); torch.cuda.empty_cache()
res = []
for rep in range(3):
    t0 = time.perf_counter()
    out = stream.run_stream_rank(lambda a, b: x[a:b], n, fs, 0, 1, 8192, 0, sub=sub)[2]
    res.append(time.perf_counter() - t0)
print("piece %.0f GiB sub %d: first pass %.3f s (%.0fx), warm %.3f / %.3f s (%.0fx) frames %d checksum %.6g" % (gib, sub, res[0], secs / res[0], res[1], res[2], secs / min(res[1:]), out.shape[0], float(np.nansum(out))))
