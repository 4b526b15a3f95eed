"""Long-stream driver (BASELINE.json configs[4]): Iterative-F0 over one long signal, time-sharded over the
GPUs of a node, ONE gather of the per-frame 12-vectors at the end.

The reference filters the whole signal sequentially through its 70-channel filterbank and then treats every
frame on its own (iterative_f0.py:54-96).  Every filter stage is stable with pole radius <= 0.999, so a rank can
start `WARMUP` samples before its first frame from zero state and lands on the sequential result to fp64
rounding (0.999^65536 ~ 3e-29) -- the same argument the front-end kernel uses for its own chunks
(csrc/mpx_if0.hip).  Frames are block-partitioned over the ranks (contiguous ranges with
a sample halo, no device-to-device exchange); the job's only collective is the all_gather of `[frames, 12]`.
Launch with ``python -m torch.distributed.run --nproc-per-node G scripts/run_stream.py ...`` (backend nccl =
RCCL), or plainly for one GPU.
"""
import json
import math
import time

import numpy as np

from .chromagram import Chromagram
from .corpus import gather_blocks, partition

WARMUP = 65536  # samples: the halo of a SUBSTITUTED compute function (CPU tests); the engine asks the library, which needs 40960 for the default chain
SEED = 20260103  # + segment id
SEGMENT_SECONDS = 0.5


def num_frames(n, frame_size):
    """frame_cutter's count (dsp/frame.py:5-14): ceil(n / frame_size), the tail frame zero-padded."""
    return -(-int(n) // int(frame_size)) if n > 0 else 0


def engine_warmup(fs, device=0, **kw):
    """Run-in samples the LIBRARY asks for with these filter-bank parameters (mpx_iterative_f0_warmup: from the slowest
    pole of the chain; 40960 for the defaults, more for channel sets that reach higher or lower).  ValueError when the
    chain cannot be cut into shards at all."""
    from .engine import get_engine
    keys = ("frame_size", "power", "channels", "zeta0", "zeta1")
    return get_engine(device).iterative_f0_warmup(fs, **{k: v for k, v in kw.items() if k in keys})[0]


def shard_window(n, frame_size, world, rank, warmup=WARMUP):
    """(f0, f1, s0, s1, skip): this rank owns frames [f0, f1) of the stream, computes on samples [s0, s1)
    and drops the first `skip` frames of what it computed (the warm-up)."""
    if int(warmup) % int(frame_size):
        raise ValueError("frame size must divide the %d-sample warm-up" % warmup)
    f0, f1 = partition(num_frames(n, frame_size), world, rank)
    return (f0, f1) + window_of(n, frame_size, f0, f1, warmup)


def window_of(n, frame_size, f0, f1, warmup=WARMUP):
    """(s0, s1, skip) for frames [f0, f1): the samples to compute on and the warm-up frames to drop."""
    if f1 <= f0:
        return 0, 0, 0
    s0 = max(0, f0 * frame_size - int(warmup))
    s1 = min(int(n), f1 * frame_size)
    return s0, s1, (f0 * frame_size - s0) // frame_size


def _engine_frames(x, fs, frame_size, device, **kw):
    from .engine import get_engine
    return get_engine(device).iterative_f0(x, fs, return_frames=True, frame_size=frame_size, **kw)[1]


def run_stream_shard(read, n, fs, rank=0, world=1, frame_size=8192, device=0, compute=None, frames=None, warmup=None,
                     **kw):
    """`read(s0, s1) -> float32[s1-s0]` hands out samples of the stream (a slice of an array, a memmap, a
    generator seeded by position).  Returns (f0, f1, frames[f1-f0, 12] float64) for this rank's frames.
    `compute(x, fs, frame_size, device, **kw) -> [F,12]` defaults to the HIP engine; tests substitute the CPU
    checker to exercise the halo logic without a GPU."""
    if warmup is None:   # the engine knows what its filter chain needs; a substituted checker gets the default chain's
        warmup = engine_warmup(fs, device, frame_size=frame_size, **kw) if compute is None else WARMUP
    compute = compute or _engine_frames
    if frames is None:
        f0, f1, s0, s1, skip = shard_window(n, frame_size, world, rank, warmup)
    else:   # an explicit frame range (a time shard inside a rank's block)
        if int(warmup) % int(frame_size):
            raise ValueError("frame size must divide the %d-sample warm-up" % warmup)
        f0, f1 = frames
        s0, s1, skip = window_of(n, frame_size, f0, f1, warmup)
    if f1 == f0:
        return f0, f1, np.zeros((0, 12), dtype=np.float64)
    x = read(s0, s1)
    if not (hasattr(x, "is_cuda") and x.is_cuda):   # a float32 tensor already on the device goes to the engine as it is
        x = np.ascontiguousarray(x, dtype=np.float32)
    if x.shape[0] != s1 - s0:
        raise ValueError("read(%d, %d) returned %d samples" % (s0, s1, x.shape[0]))
    frames = np.asarray(compute(x, fs, frame_size, device, **kw), dtype=np.float64)
    return f0, f1, frames[skip:skip + (f1 - f0)]


_ENGINES = {}


def _engine_frames_on(j):
    """compute function bound to the j-th context of a device (0 = the process-wide engine)."""
    def compute(x, fs, frame_size, device, **kw):
        from .engine import Engine, get_engine
        key = (int(device), j)
        if key not in _ENGINES:
            _ENGINES[key] = get_engine(device) if j == 0 else Engine(device)
        return _ENGINES[key].iterative_f0(x, fs, return_frames=True, frame_size=frame_size, **kw)[1]
    return compute


PIECE_BYTES = 32 << 30   # front-end output of one piece of an explicit `sub` > 1 run (560 B per sample at 70 channels)
STREAM_WORKSPACE_BYTES = 12 << 30  # hand-off buffer of ONE engine call over a rank's whole share (MPX_OPT_IF0_WORKSPACE_BYTES):
                                   # the library advances every chunk in time slices of whole frames and carries the filter
                                   # state over, so an hour of 44.1 kHz audio is still ONE front-end launch geometry of 923
                                   # long chunks (a run-in of 24 % per chunk) in 7 slices of 3 frames per chunk, and its
                                   # workspaces are 13 GB instead of round 3's 90 GB, whose first hipMalloc cost 2-4 s on a
                                   # device that had been used before.  Measured (scripts/dev/if0_hour_caps.py, MI355X):
                                   # 12 GiB 0.131 s warm, 8 GiB 0.135 s (11 slices of 2 frames: the summary-spectrum kernel's
                                   # eighth round a fifth full), 4.5 GiB 0.139 s, one 83 GiB piece 0.123 s


def _workspace_cap(device, want):
    """`want` bytes, but never more than a third of the device memory that is free right now (>= 256 MiB)."""
    try:
        import torch
        if torch.cuda.is_available():
            free, _ = torch.cuda.mem_get_info(int(device))
            want = min(int(want), int(free) // 3)
    except Exception:
        pass
    return max(int(want), 256 << 20)


def _engine_frames_capped(j, cap):
    """compute function of context j with the Iterative-F0 workspace cap set for the call (and restored afterwards: the
    process-wide engine also serves clip batches, which the cap would cut into more passes)."""
    inner = _engine_frames_on(j)

    def compute(x, fs, frame_size, device, **kw):
        from .engine import get_engine
        eng = _ENGINES.get((int(device), j)) or (get_engine(device) if j == 0 else None)
        if eng is None:
            return inner(x, fs, frame_size, device, **kw)
        before = eng.get_option("if0_workspace_bytes")
        eng.set_option("if0_workspace_bytes", _workspace_cap(device, cap))
        try:
            return inner(x, fs, frame_size, device, **kw)
        finally:
            eng.set_option("if0_workspace_bytes", before)
    return compute


def run_stream_rank(read, n, fs, rank=0, world=1, frame_size=8192, device=0, sub=None, channels=70,
                    workspace_bytes=None, **kw):
    """This rank's frames.  sub=None or 1: ONE engine call over the rank's whole share, whatever its length -- the library
    bounds its own workspaces (time slices, include/mpx.h MPX_OPT_IF0_WORKSPACE_BYTES; `workspace_bytes`, default
    STREAM_WORKSPACE_BYTES, at most a third of the free device memory).  sub > 1: that many time shards IN FLIGHT on the
    same GPU (one context and one host thread each; ctypes releases the GIL), the rank's block of frames partitioned once
    more with the same halo logic as between ranks.  Returns (f0, f1, frames[f1-f0, 12])."""
    import threading
    warmup = engine_warmup(fs, device, frame_size=frame_size, channels=channels, **kw)
    F0, F1 = partition(num_frames(n, frame_size), world, rank)
    if sub is None or sub == 1:
        cap = STREAM_WORKSPACE_BYTES if workspace_bytes is None else workspace_bytes
        return run_stream_shard(read, n, fs, rank, world, frame_size, device, compute=_engine_frames_capped(0, cap),
                                frames=(F0, F1), warmup=warmup, channels=channels, **kw)
    # pieces: a multiple of `sub`, each small enough for PIECE_BYTES of front-end output (a two-hour stream would
    # otherwise ask one context for 180 GB, and the library refuses beyond 96 GiB per call)
    per_frame = int(frame_size) * int(channels) * 8
    rounds = max(1, -(-((F1 - F0) * per_frame) // (sub * PIECE_BYTES)))
    pieces = sub * rounds
    parts = [None] * pieces
    failed = []

    def work(j):   # context j takes pieces j, j + sub, j + 2 sub, ...
        try:
            for q in range(j, pieces, sub):
                a, b = partition(F1 - F0, pieces, q)
                parts[q] = run_stream_shard(read, n, fs, rank, world, frame_size, device,
                                            compute=_engine_frames_on(j), frames=(F0 + a, F0 + b), warmup=warmup,
                                            channels=channels, **kw)
        except BaseException as exc:   # a side thread's exception is re-raised by the caller below
            failed.append(exc)

    threads = [threading.Thread(target=work, args=(j,)) for j in range(1, sub)]
    for t in threads:
        t.start()
    work(0)
    for t in threads:
        t.join()
    if failed:
        raise failed[0]
    if any(p is None for p in parts):
        raise RuntimeError("a time shard failed")
    return parts[0][0], parts[-1][1], np.concatenate([p[2] for p in parts], axis=0)


def gather_frames(block, total_frames, world, rank, device=None, force=False):
    """all_gather of the per-rank `[frames, 12]` blocks -> `[total_frames, 12]` on every rank."""
    return gather_blocks(block[:, None, :], total_frames, world, rank, device, force=force)[:, 0, :]


def chroma_of(frames):
    """The reference's accumulation (iterative_f0.py:87-96 via Chromagram.__add__): frame by frame, in order."""
    acc = np.zeros(12, dtype=np.float64)
    for row in frames:
        acc = acc + row
    return Chromagram(acc)


# ------------------------------------------------------------------ synthetic stream (SURVEY.md 8d, M-stream)
def segment_notes(seg):
    rng = np.random.default_rng(SEED + int(seg))
    k = int(rng.integers(3, 7))
    return [(int(rng.integers(36, 85)), float(rng.uniform(0.0, 2.0 * math.pi))) for _ in range(k)]


def synth_stream(s0, s1, fs, device=None):
    """Samples [s0, s1) of an endless stream: every 0.5 s segment holds 3-6 notes (8 harmonics decaying by 0.7)
    at MIDI pitches 36..84 plus white noise at -40 dBFS, scaled by a fixed 0.05.  A function of the sample
    position only (segment id -> seed), so every rank synthesises exactly its own window."""
    import torch
    dev = torch.device(device) if device is not None else torch.device("cpu")
    seg_len = int(round(SEGMENT_SECONDS * fs))
    out = torch.empty(int(s1 - s0), dtype=torch.float32, device=dev)
    for seg in range(int(s0) // seg_len, (int(s1) + seg_len - 1) // seg_len):
        a, b = max(int(s0), seg * seg_len), min(int(s1), (seg + 1) * seg_len)
        t = torch.arange(a, b, dtype=torch.float64, device=dev) / float(fs)
        y = torch.zeros(b - a, dtype=torch.float64, device=dev)
        for midi, ph in segment_notes(seg):
            f0 = 440.0 * 2.0 ** ((midi - 69) / 12.0)
            for h in range(1, 9):
                if f0 * h < fs / 2:
                    y += (0.7 ** (h - 1)) * torch.sin(2.0 * math.pi * f0 * h * t + ph * h)
        g = torch.Generator(device="cpu")
        g.manual_seed(SEED + 7919 * seg)
        noise = torch.randn(seg_len, generator=g, dtype=torch.float64)[a - seg * seg_len:b - seg * seg_len].to(dev)
        out[a - int(s0):b - int(s0)] = (0.05 * y + 0.003 * noise).to(torch.float32)
    return out


def main(argv=None, compute=None, device="cuda", backend="nccl"):
    """`compute`, `device`, `backend`: injection points for the CPU tests (a checker instead of the engine, CPU tensors,
    gloo): the command line never sets them -- without a GPU and the HIP library this driver fails, it has no fallback."""
    import argparse
    import os
    ap = argparse.ArgumentParser(description="Iterative-F0 over one long synthetic stream, time-sharded over the GPUs of a node")
    ap.add_argument("--seconds", type=float, default=3600.0)
    ap.add_argument("--fs", type=int, default=44100)
    ap.add_argument("--frame-size", type=int, default=8192)
    ap.add_argument("--shards-per-gpu", type=int, default=0,
                    help="time shards in flight on each GPU, own context each; 0 (default): one engine call over the rank's "
                         "share when its front-end output fits 90 GiB, else 3 (1 h @44.1 kHz on one MI355X, round 3: "
                         "0.13-0.14 s in one call, 0.145-0.15 s with 3 shards)")
    ap.add_argument("--force-collective", action="store_true",
                    help="create the process group and run the closing all_gather even with ONE rank (RCCL on a one-GPU box)")
    args = ap.parse_args(argv)
    from . import launch
    rank, world, local = launch.rank_world_local()
    import torch
    on_gpu = device == "cuda"
    dev = torch.device("cuda", local) if on_gpu else torch.device(device)
    use_dist = launch.wants_collective(world, args.force_collective)
    if use_dist:   # before anything else touches the GPU: the communicator is bound to the device here
        dist = launch.init_group(backend, dev if on_gpu else None)
    if on_gpu:
        torch.cuda.set_device(dev)
    n = int(round(args.seconds * args.fs))
    total_frames = num_frames(n, args.frame_size)

    def dev_sync():
        if on_gpu:
            torch.cuda.synchronize()

    def read(s0, s1):   # stays in HBM: the engine takes device memory (include/mpx.h, "where the samples live")
        return synth_stream(s0, s1, args.fs, dev)

    if compute is None:
        for j in range(max(1, args.shards_per_gpu or 3)):   # plans, tables, clocks -- of every context
            _engine_frames_on(j)(read(0, min(n, 4 * args.frame_size)), args.fs, args.frame_size, local)
    if use_dist:
        dist.barrier()
    dev_sync()
    t0 = time.perf_counter()
    warm = engine_warmup(args.fs, local, frame_size=args.frame_size) if compute is None else WARMUP
    f0, f1, s0, s1, _ = shard_window(n, args.frame_size, world, rank, warm)
    x = read(s0, s1)
    dev_sync()   # the synthesis is asynchronous; it is not part of the measured path
    if on_gpu:
        torch.cuda.empty_cache()  # the synthesis' cached blocks: with them in place the engine's first hipMalloc of its workspace takes ~1 s
    t1 = time.perf_counter()
    t_synth = t1 - t0

    def compute_block():
        if compute is not None:    # the CPU tests' checker: one shard per rank, same halo logic
            return run_stream_shard(lambda a, b: x.numpy(), n, args.fs, rank, world, args.frame_size, local,
                                    compute=compute)[2]
        if args.shards_per_gpu != 1:
            return run_stream_rank(lambda a, b: x[a - s0:b - s0], n, args.fs, rank, world, args.frame_size, local,
                                   sub=args.shards_per_gpu or None)[2]
        return run_stream_shard(lambda a, b: x, n, args.fs, rank, world, args.frame_size, local)[2]

    # twice: the first pass grows the contexts' workspaces (tens of GB of hipMalloc: 0.1 ... 3 s, whatever state the
    # driver is in), the second is what a service that processes one stream after the other sees
    block = compute_block()
    t_cold = time.perf_counter() - t1
    if use_dist:
        dist.barrier()
    t1 = time.perf_counter()
    block = compute_block()
    t2 = time.perf_counter()
    frames = gather_frames(block, total_frames, world, rank, dev if (use_dist and on_gpu) else None, force=use_dist)
    spent = torch.tensor([t_synth, t2 - t1, t_cold], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(spent, op=dist.ReduceOp.MAX)
    if rank == 0:
        c = chroma_of(frames)
        synth_s, compute_s, cold_s = (float(v) for v in spent.cpu())
        print(json.dumps({"workload": "Iterative-F0, %.0f s stream @%d Hz (BASELINE configs[4])" % (args.seconds, args.fs),
                          "n_gpus": world, "collective": ("%s all_gather over %d rank(s)" % (backend, world)) if use_dist else None,
                          "frames": total_frames, "frames_per_rank": f1 - f0,
                          "synthesis_seconds_rank_max": synth_s, "compute_seconds_rank_max": compute_s,
                          "x_realtime_compute": args.seconds / compute_s if compute_s > 0 else None,
                          "compute_seconds_first_pass": cold_s, "shards_in_flight_per_gpu": args.shards_per_gpu or "auto (one call when the share fits 90 GiB of front-end output, else 3)",
                          "chroma": repr(c), "key": c.key()}))
    if use_dist:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    main()
