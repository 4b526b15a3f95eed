"""Summary-spectrum kernel time per frame when every frame of a clip is whole (clips of 5 x 8192 samples) against the corpus'
two-second clips (44 100 samples: five whole frames and one of 3140 samples, which takes the kernel's select-and-no-prefetch body)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import chord_detection_amd as cd
from chord_detection_amd import corpus
fs, n = 22050, 4096
block = corpus.synth_block(n, fs, 2.0, 1024, 0, 1, synth_device="cuda:0")
clips = block[0][1]
torch.cuda.synchronize()
eng = cd.Engine(0)
for L in (40960, 44100, 49152):
    x = clips[:, :L].contiguous() if L <= clips.shape[1] else torch.cat([clips, clips[:, :L - clips.shape[1]]], dim=1).contiguous()
    frames = n * -(-L // 8192)
    for rep in range(3):
        eng.profile_begin()
        t0 = time.perf_counter()
        eng.iterative_f0_batch(x, fs)
        wall = time.perf_counter() - t0
        prof = eng.profile_end()
    ks = {k: round(v[1], 2) for k, v in prof.items()}
    print("clips of %d samples (%d frames): wall %.1f ms, %s; spectra %.3f us per frame, front end %.3f ns per sample" % (
        L, frames, 1e3 * wall, ks, 1e3 * ks["if0_spectrum_kernel"] / frames, 1e6 * ks["if0_frontend_kernel"] / (n * L)), flush=True)
