import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
import chord_detection_amd as cd
from chord_detection_amd import stream
fs, secs, nf = 44100, 3600.0, 8192
n = int(round(secs * fs))
x = stream.synth_stream(0, n, fs, torch.device("cuda", 0))
torch.cuda.synchronize(); torch.cuda.empty_cache()
for sub in (1, 2, 3, 4, 2):
    for rep in range(2):
        t0 = time.perf_counter()
        blk = stream.run_stream_rank(lambda a, b: x[a:b], n, fs, 0, 1, nf, 0, sub=sub)[2]
        w = time.perf_counter() - t0
    print("sub", sub, "wall %.4f s -> %.0f x real time" % (w, secs / w), blk.shape)
