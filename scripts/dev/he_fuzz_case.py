import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import harmonic_energy as o_he
rng = np.random.default_rng(551)
def signal(n, fs):
    t = np.arange(n) / fs
    x = np.zeros(n)
    for _ in range(int(rng.integers(1, 5))):
        f0 = 440.0 * 2.0 ** ((int(rng.integers(30, 90)) - 69) / 12.0)
        for h in range(1, 7):
            if f0 * h < fs / 2:
                x += 0.7 ** h * np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 6.28))
    x += rng.choice([0.0, 1e-3, 0.1]) * rng.standard_normal(n)
    return (rng.uniform(0.01, 1.0) * x / max(np.abs(x).max(), 1e-9)).astype(np.float32)
for case in range(215):
    fs = int(rng.choice([16000, 22050, 44100, 48000]))
    N = int(rng.choice([1024, 2048, 4096, 8192, 16384, 600, 1000, 1023, 2227, 3000, 4095, 5000, 6500, 8000, 12000, 20000, 32768]))
    hop = min(N, int(rng.choice([N, N // 2, N // 4, 1000])))
    n = int(rng.choice([N - 1, N, N + 1, 3 * N + 5, 20 * hop + N]))
    kw = dict(num_harmonic=int(rng.integers(1, 4)), num_octave=int(rng.integers(1, 4)), num_bins=int(rng.integers(0, 4)))
    x = signal(n, fs)
print(case, fs, N, hop, n, kw)
np.save("gpurun_out/he_case_x.npy", x)
want = o_he.he_frames(x.astype(np.float64), fs, N, hop, **kw)
# long-double reference of the same rows: direct DFT at the window bins
import inspect
print(want.shape, want.max(), want.min())
np.save("gpurun_out/he_case_want.npy", want)
if "--gpu" in sys.argv:
    import chord_detection_amd as cd
    eng = cd.get_engine(0)
    tot, per = eng.harmonic_energy(x, fs, N, hop, return_frames=True, **kw)
    rel = np.abs(per - want) / np.maximum(np.abs(want), 1e-300)
    i = np.unravel_index(np.argmax(rel), rel.shape)
    print("max rel", rel.max(), "at", i, "got", per[i], "want", want[i], "row max", want[i[0]].max())
    np.save("gpurun_out/he_case_per.npy", per)
