"""1 h stream through ONE engine call at several workspace caps (time slices): first call of a fresh context (cold: its
workspaces are allocated inside) and the third call (warm), per-kernel times of the warm call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import chord_detection_amd as cd
from chord_detection_amd import stream
fs, secs = 44100, 3600
CH = int(os.environ.get("IF0_CHANNELS", "70"))   # e.g. 64: no waves of leftover channels
x = stream.synth_stream(0, secs * fs, fs, "cuda:0")
torch.cuda.synchronize(); torch.cuda.empty_cache()
ref = None
for arg in (sys.argv[1:] or ["4", "8", "16", "32", "90"]):   # "12": one buffer; "24o": MPX_IF0_OVERLAP=1 (development library: two buffers of half the cap, spectra beside the next front end)
    gib, ov = float(arg.rstrip("o")), int(arg.endswith("o"))
    eng = cd.Engine(0)
    os.environ["MPX_IF0_OVERLAP"] = str(ov)
    if hasattr(eng.lib, "mpx_set_option"):
        eng.set_option("if0_workspace_bytes", int(gib * (1 << 30)))
    d_frames = torch.empty((stream.num_frames(x.numel(), 8192), 12), dtype=torch.float64, device="cuda:0")
    walls = []
    for rep in range(4):
        if rep == 3:
            eng.profile_begin()
        t0 = time.perf_counter()
        eng.iterative_f0_dev(x.data_ptr(), x.numel(), fs, d_frames.data_ptr(), None, frame_size=8192, channels=CH)
        eng.synchronize()
        walls.append(time.perf_counter() - t0)
    prof = eng.profile_end()
    r = d_frames.cpu().numpy()
    if ref is None:
        ref = r
    print("cap %5.1f GiB%s: cold %.3f s (%.0fx)  warm %.1f / %.1f ms (%.0fx)  %s  equal to first cap: %s" % (
        gib, " two buffers" if ov else "", walls[0], secs / walls[0], 1e3 * walls[1], 1e3 * walls[2], secs / min(walls[1:3]),
        {k: (v[0], round(v[1], 1)) for k, v in prof.items()}, bool(np.array_equal(r, ref))), flush=True)
    eng.close()
    del d_frames
    torch.cuda.empty_cache()
