"""TESTS ONLY: a stand-in for torch.cuda + the HIP engine, so that the multi-rank control flow of bench.py (barriers,
the one all_gather, max-over-ranks timing, the sharded workloads, rank 0's JSON line) runs on CPU over gloo with
world_size 2.  Selected by the test through MPX_BENCH_STUB=tests.bench_stub; nothing in the product imports it.
The "engine" here computes the Harmonic-Energy rows with the oracle and cheap deterministic fakes for the rest: the
numbers mean nothing, the plumbing is what is exercised."""
import ctypes
import time

import numpy as np


def configure(bench):
    """Shrink the bench to toy sizes."""
    bench.FRAMES = 8
    bench.PREHEAT_MS = 0
    bench.NSIG = 2
    bench.CFG.update(esacf_clips=3, esacf_fs=22050, esacf_clip_seconds=0.25, corpus_clips_per_gpu=3, corpus_fs=22050,
                     stream_seconds=9.0, stream_fs=22050, if0_frame=8192, he_default_clips=2, he_default_fs=22050,
                     he_default_frame=1024)


def _f32(ptr, n):
    return np.ctypeslib.as_array((ctypes.c_float * int(n)).from_address(int(ptr)))


def _f64(ptr, n):
    return np.ctypeslib.as_array((ctypes.c_double * int(n)).from_address(int(ptr)))


def _fake_rows(x, frame, hop):
    """[F, 12]: a cheap deterministic function of the frames (not ESACF; see the module docstring)."""
    n = x.shape[0]
    nf = max(0, -(-n // frame)) if hop == frame else (1 if n <= frame else 1 + -(-(n - frame) // hop))
    out = np.zeros((nf, 12))
    for f in range(nf):
        seg = np.abs(np.asarray(x[f * hop:f * hop + frame], dtype=np.float64))
        for b in range(12):
            out[f, b] = seg[b::12].sum()
    return out


class Engine:
    def __init__(self):
        self._t0 = 0.0
        self._prof = None

    def synchronize(self):
        pass

    def num_frames(self, n, frame, hop=None):
        hop = hop or frame
        return -(-n // frame) if hop == frame else (1 if n <= frame else 1 + -(-(n - frame) // hop))

    def timer_begin(self, stream=None):
        self._t0 = time.perf_counter()

    def timer_end(self, stream=None):
        return 1e3 * (time.perf_counter() - self._t0)

    def profile_begin(self):
        self._prof = {}

    def profile_end(self):
        p, self._prof = self._prof or {}, None
        return p or {"sacf_kernel": (1, 1.0), "bandsplit_kernel": (1, 0.5)}

    def harmonic_energy_dev(self, d_signal, n, fs, frame, hop, d_frames, d_sum, stream=None, **kw):
        from oracle import harmonic_energy as o_he
        rows = o_he.he_frames(_f32(d_signal, n), fs, frame, hop)
        if d_frames:
            _f64(d_frames, rows.size)[:] = rows.reshape(-1)
        if d_sum:
            acc = np.zeros(12)
            for r in rows:
                acc = acc + r
            _f64(d_sum, 12)[:] = acc
        if self._prof is not None:
            self._prof["he_kernel"] = (1, 0.05)

    def harmonic_energy_batch(self, clips, fs, frame=8192, hop=None, **kw):
        from oracle import harmonic_energy as o_he
        clips = clips.numpy() if hasattr(clips, "numpy") else np.asarray(clips)
        return np.stack([o_he.he_frames(c, fs, frame, hop or frame).sum(axis=0) for c in clips])

    def esacf_dev(self, d_signal, n, fs, frame, hop, d_frames, d_sum, stream=None, **kw):
        rows = _fake_rows(_f32(d_signal, n), frame, hop)
        if d_frames:
            _f64(d_frames, rows.size)[:] = rows.reshape(-1)
        if d_sum:
            _f64(d_sum, 12)[:] = rows.sum(axis=0)

    def esacf_batch(self, clips, fs, frame, hop=None, **kw):
        clips = clips.numpy() if hasattr(clips, "numpy") else np.asarray(clips)
        return np.stack([_fake_rows(c, frame, hop or frame).sum(axis=0) for c in clips])


def corpus_compute(method, clips, fs, device):
    clips = np.asarray(clips)
    if method == 2:
        from oracle import harmonic_energy as o_he
        return np.stack([o_he.he_compute(c, fs) for c in clips])
    return np.stack([_fake_rows(c, 1024 * method, 1024 * method).sum(axis=0) + method for c in clips])


def stream_compute(x, fs, frame_size, device, **kw):
    return _fake_rows(np.asarray(x), frame_size, frame_size) + 1.0
