import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import chord_detection_amd as cd
eng = cd.Engine(0); dev = torch.device("cuda", 0)
for fs, N in ((44100, 4096), (192000, 8908)):
    F = 4096; n = F * N
    rng = np.random.default_rng(fs)
    t = np.arange(8 * N) / float(fs)
    base = sum(0.6 ** h * np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 6)) for f0 in (146.83, 220.0, 277.18) for h in range(1, 6))
    base = (0.25 * base + 0.004 * rng.standard_normal(t.shape[0])).astype(np.float32)
    x = torch.from_numpy(np.tile(base, F // 8)).to(dev)
    d_frames = torch.zeros((F, 12), dtype=torch.float64, device=dev); d_sum = torch.zeros(12, dtype=torch.float64, device=dev)
    for _ in range(2):
        eng.esacf_dev(x.data_ptr(), n, fs, N, N, d_frames.data_ptr(), d_sum.data_ptr()); eng.synchronize()
    eng.profile_begin()
    eng.esacf_dev(x.data_ptr(), n, fs, N, N, d_frames.data_ptr(), d_sum.data_ptr()); eng.synchronize()
    prof = eng.profile_end()
    print(fs, N, {k: round(v[1], 3) for k, v in prof.items() if "fit" in k})
