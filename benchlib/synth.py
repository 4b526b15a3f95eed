"""Synthetic inputs of the benchmark (SURVEY.md 8d recipes)."""
from . import config as K


def synth_signal(seed, frames=None):
    """SURVEY.md 8(d) M-HE: decaying-harmonic notes at random MIDI 36-84, 0.5 s each,
    + white noise at -40 dBFS, peak 0.9.  float32, (frames-1)*hop + N samples."""
    import numpy as np
    frames = K.FRAMES if frames is None else frames
    n = (frames - 1) * K.HOP + K.N_FFT
    rng = np.random.default_rng(seed)
    seg = K.FS // 2
    x = np.zeros(n, dtype=np.float64)
    t = np.arange(seg) / K.FS
    for s0 in range(0, n, seg):
        m = min(seg, n - s0)
        acc = np.zeros(m)
        for _ in range(int(rng.integers(3, 7))):
            f0 = 440.0 * 2.0 ** ((int(rng.integers(36, 85)) - 69) / 12.0)
            ph = rng.uniform(0, 2 * np.pi)
            for h in range(1, 5):
                acc += (0.5 ** (h - 1)) * np.sin(2 * np.pi * f0 * h * t[:m] + ph)
        x[s0:s0 + m] = acc
    x /= np.max(np.abs(x))
    x += 0.01 * rng.standard_normal(n)
    x *= 0.9 / np.max(np.abs(x))
    return x.astype(np.float32)


def synth_clips_numpy(count, fs, seconds):
    """float32 [count, fs*seconds] polyphonic clips in the recipe of the corpus driver (2-4 notes of 8 harmonics decaying
    by 0.7, noise at 0.003, peak 0.9), NumPy only: the CPU legs run before torch or HIP are loaded."""
    import numpy as np
    n = int(round(fs * seconds))
    t = np.arange(n) / fs
    out = np.zeros((count, n), dtype=np.float32)
    for c in range(count):
        rng = np.random.default_rng(20260102 + c)
        y = np.zeros(n)
        for _ in range(int(rng.integers(2, 5))):
            f0 = 440.0 * 2.0 ** ((int(rng.integers(36, 85)) - 69) / 12.0)
            ph = rng.uniform(0, 2 * np.pi)
            for h in range(1, 9):
                if f0 * h < fs / 2:
                    y += (0.7 ** (h - 1)) * np.sin(2 * np.pi * f0 * h * t + ph * h)
        y += 0.003 * rng.standard_normal(n)
        out[c] = (0.9 * y / np.max(np.abs(y))).astype(np.float32)
    return out


def synth_signal_device(seed, dev, frames=None):
    """The same recipe as synth_signal, evaluated with torch on `dev` (a few tensor ops instead of ~9000 NumPy ones per
    signal: the bench rotates over NSIG of them).  Not sample-identical to synth_signal (other random streams)."""
    import numpy as np
    import torch
    frames = K.FRAMES if frames is None else frames
    n = (frames - 1) * K.HOP + K.N_FFT
    rng = np.random.default_rng(seed)
    seg = K.FS // 2
    nseg = -(-n // seg)
    tab = np.zeros((nseg, 24, 3))      # [segment, 6 notes x 4 harmonics, (angular frequency, phase, amplitude)]
    for s in range(nseg):
        for k in range(int(rng.integers(3, 7))):
            f0 = 440.0 * 2.0 ** ((int(rng.integers(36, 85)) - 69) / 12.0)
            ph = rng.uniform(0, 2 * np.pi)
            for h in range(1, 5):
                tab[s, 4 * k + h - 1] = (2 * np.pi * f0 * h, ph, 0.5 ** (h - 1))
    t = torch.arange(seg, dtype=torch.float64, device=dev) / K.FS
    tab_t = torch.from_numpy(tab).to(dev)
    x = torch.empty(nseg * seg, dtype=torch.float64, device=dev)
    for s0 in range(0, nseg, 32):
        w = tab_t[s0:s0 + 32]
        x[s0 * seg:(s0 + w.shape[0]) * seg] = (w[:, :, 2:3] * torch.sin(w[:, :, 0:1] * t + w[:, :, 1:2])).sum(dim=1).reshape(-1)
    x = x[:n]
    x /= x.abs().max()
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    x += 0.01 * torch.randn(n, generator=g, dtype=torch.float64).to(dev)
    x *= 0.9 / x.abs().max()
    return x.to(torch.float32).contiguous()

