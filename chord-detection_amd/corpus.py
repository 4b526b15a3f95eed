"""Corpus driver (BASELINE.json configs[3]): all four methods over a synthetic corpus of short clips,
clip-sharded over the GPUs of one node, ONE gather of the per-clip 12-vectors at the end.

The reference has no corpus runner -- its CLI handles one file (chord_detect.py:45-63) and its tests loop
over a directory in Python (tests/test.py:18-33).  This is that loop, batched: every rank takes a
contiguous block of clip ids (clip-granular block partition, no data-path
collective), synthesises its clips chunk by chunk (never stored), runs the chunk through each method's
``compute_batch`` (the batch entry points of the C ABI) and keeps one ``[12]`` vector per clip and method.
Launch with ``python -m torch.distributed.run --nproc-per-node G scripts/run_corpus.py ...`` for G GPUs
(backend nccl = RCCL), or plainly for one.
"""
import json
import math
import os
import time

import numpy as np

from .chromagram import Chromagram
from .multipitch import METHODS

SEED = 20260102  # + clip id (SURVEY.md section 8d, M-ESACF / M-corpus)


def partition(n_clips, world, rank):
    """Contiguous block of clip ids owned by `rank`: [lo, hi).  Blocks differ by at most one clip."""
    if not (0 <= rank < world):
        raise ValueError("rank outside world")
    base, rem = divmod(int(n_clips), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def clip_notes(clip_id):
    """2-4 random MIDI pitches 36..84 with a phase each, seeded by the clip id."""
    rng = np.random.default_rng(SEED + int(clip_id))
    k = int(rng.integers(2, 5))
    return [(int(rng.integers(36, 85)), float(rng.uniform(0.0, 2.0 * math.pi))) for _ in range(k)]


def synth_chunk(clip_ids, fs, seconds, device=None):
    """float32 [len(clip_ids), fs*seconds] polyphonic clips: every note has 8 harmonics decaying by 0.7,
    white noise at 0.003, peak-normalised to 0.9.  Tones are summed in float64 with torch on `device`
    (CPU when None).  The noise is a counter-based hash of (clip id, sample index) -- no generator state -- so a clip is
    the same samples whatever chunk, rank or shard it is synthesised in (a sharded corpus equals the unsharded one exactly,
    for a given device type)."""
    import torch
    dev = torch.device(device) if device is not None else torch.device("cpu")
    n = int(round(fs * seconds))
    t = torch.arange(n, dtype=torch.float64, device=dev) / float(fs)
    out = torch.zeros((len(clip_ids), n), dtype=torch.float64, device=dev)
    # partial table [clips, 4 notes x 8 harmonics]: angular frequency, phase, amplitude (0 = unused slot)
    tab = np.zeros((len(clip_ids), 32, 3))
    for row, cid in enumerate(clip_ids):
        for k, (midi, ph) in enumerate(clip_notes(cid)):
            f0 = 440.0 * 2.0 ** ((midi - 69) / 12.0)
            for h in range(1, 9):
                if f0 * h < fs / 2:
                    tab[row, k * 8 + h - 1] = (2.0 * math.pi * f0 * h, ph * h, 0.7 ** (h - 1))
    tab_t = torch.from_numpy(tab).to(dev)
    for r0 in range(0, len(clip_ids), 32):  # [32 clips, 32 partials, n] float64 at a time
        w = tab_t[r0:r0 + 32]
        out[r0:r0 + 32] = (w[:, :, 2:3] * torch.sin(w[:, :, 0:1] * t + w[:, :, 1:2])).sum(dim=1)
    if len(clip_ids):
        out += 0.003 * _clip_noise(torch.tensor([int(c) for c in clip_ids], dtype=torch.int32, device=dev), n)
    out *= 0.9 / out.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
    return out.to(torch.float32)


def _clip_noise(ids, n):
    """Approximately normal noise [len(ids), n] (variance 1) as a pure function of (clip id, sample index): one 32-bit
    integer hash per sample, the sum of its four bytes (Irwin-Hall, 4 terms).  int32 arithmetic wraps identically on every
    device; one hash round instead of a generator keeps the synthesis a small fraction of a chunk's GPU time."""
    import torch

    def c32(v):   # a 32-bit constant as the signed value int32 tensors hold
        v &= 0xFFFFFFFF
        return v - (1 << 32) if v & 0x80000000 else v

    j = torch.arange(n, dtype=torch.int32, device=ids.device)
    h = (ids.to(torch.int32)[:, None] + c32(SEED)) * c32(0x9E3779B1) + j[None, :] * c32(0x85EBCA77)
    h = h ^ ((h >> 15) & 0x1FFFF)          # logical shifts: mask off the sign extension
    h = h * c32(0x2C1B3C6D)
    h = h ^ ((h >> 12) & 0xFFFFF)
    h = h * c32(0x297A2D39)
    h = h ^ ((h >> 15) & 0x1FFFF)
    b = (h & 0xFF) + ((h >> 8) & 0xFF) + ((h >> 16) & 0xFF) + ((h >> 24) & 0xFF)
    return (b.to(torch.float64) - 510.0) * (1.0 / 147.80054127099806)   # 4 uniform bytes: mean 510, variance 4 (256^2 - 1) / 12


def _engine_compute(method, clips, fs, device, note_names="unicode"):
    kw = {} if method == 2 else {"note_names": note_names}   # method 2 never spells a note (harmonic_energy.py:66-67)
    return np.stack([c.as_array() for c in METHODS[method].compute_batch(clips, fs, device=device, **kw)])


_SECOND_ENGINE = {}


LAST_GATE_SECONDS = (0.0, 0)   # how long the last gated side thread waited, and how many polls that took (diagnostics)
SIDE_THREADS_WAIT = True   # False: the side threads start at once (rounds 3-5; scripts/dev/corpus_order.py A/B)


def _start_side(th, main_engine, main_done):
    """Start a side thread (Iterative-F0 on a context of its own) so that its kernels reach the GPU BEHIND the main thread's
    first ones.  Iterative-F0's front end fills every SIMD's registers with long-lived one-wave workgroups, and ESACF's
    multi-wave workgroups then find no CU with room until it has drained: whenever the front end won the race to the GPU, ESACF
    ran AFTER the 131 ms of Iterative-F0 instead of beside it -- 160 ms per 4096-clip group instead of 150
    (profiles/r6/corpus_head_start.txt: every one of four fresh processes, and three of this project's eleven evidence runs).
    It won whenever something held the main thread's call up: in those runs the library's hipFree of an outgrown workspace, a
    device-wide wait (csrc/mpx_api.hip `ensure`: such blocks are retired now, profiles/r6/corpus_retire_cap_ab.txt).  The gate
    makes the order independent of such stalls: the thread waits until the main context has enqueued a kernel
    (mpx_launch_count moves), the main thread is through with its methods, or 50 ms have passed.
    (hipStreamQuery on the main context's stream does not do: it waits behind the main thread's hipStreamSynchronize and
    answers "idle" when the call is over -- profiles/r6/corpus_gate_stream_query.txt.)"""
    if not SIDE_THREADS_WAIT or main_engine is None:
        th.start()
        return
    real_run = th.run
    try:
        count0 = main_engine.launch_count()
    except Exception:   # a library without the counter: start as before
        th.start()
        return

    def gated():
        global LAST_GATE_SECONDS
        t_in = time.perf_counter()
        t_end = t_in + 0.05
        polls = 0
        while not main_done.is_set() and time.perf_counter() < t_end and main_engine.launch_count() == count0:
            time.sleep(0.0001)
            polls += 1
        LAST_GATE_SECONDS = (time.perf_counter() - t_in, polls)
        real_run()

    th.run = gated
    th.start()


LAST_SYNTH_SECONDS = 0.0   # input synthesis inside the last run_corpus call of this process (reported next to the wall clock)


def _second_engine(device):
    """A second context (own stream, own workspaces) on the same GPU: Iterative-F0 runs there next to the other methods."""
    from .engine import Engine
    if device not in _SECOND_ENGINE:   # key: the device, or (device, n) for the n-th context on it
        _SECOND_ENGINE[device] = Engine(device[0] if isinstance(device, tuple) else device)
    return _SECOND_ENGINE[device]


def synth_block(n_clips, fs=22050, seconds=2.0, chunk=1024, rank=0, world=1, synth_device=None, group=4):
    """This rank's clips, synthesised up front: {first clip id of a group of chunks: (ids, clips)} for run_corpus(resident=...)
    -- a corpus that is already in HBM when the clock starts (bench.py; 4096 two-second clips are 0.7 GB)."""
    lo, hi = partition(n_clips, world, rank)
    out = {}
    chunk = chunk * max(1, int(group))   # see run_corpus
    for c0 in range(lo, hi, chunk):
        ids = list(range(c0, min(c0 + chunk, hi)))
        out[c0] = (ids, synth_chunk(ids, fs, seconds, synth_device))
    if synth_device is not None and str(synth_device).startswith("cuda"):
        import torch
        torch.cuda.synchronize()
    return out


def run_corpus(n_clips, methods=(1, 2, 3, 4), fs=22050, seconds=2.0, chunk=1024, rank=0, world=1, device=0,
               compute=None, synth_device=None, overlap=True, note_names="unicode", resident=None, group=None):
    """Process this rank's block.  Returns (lo, hi, chroma[hi-lo, len(methods), 12] float64, seconds per method).
    `compute(method, clips, fs, device) -> [n,12]` defaults to the HIP engine's batch entry points; tests
    substitute a CPU function to exercise the sharding logic without a GPU.  `resident`: the groups of synth_block()
    (same n_clips / chunk / group / rank / world) -- nothing is synthesised inside the call then.
    `chunk` clips are what Iterative-F0 takes per engine call (its front-end output is 27 MB per two-second clip); the other
    methods take `group` chunks at once: ESACF ends every call on the latency of its last runaway gaussian fits (5 of the 13 ms
    a 1024-clip call takes: four such calls cost 53 ms where one call over 4096 clips costs 22), and no method waits for
    another at a chunk boundary any more.  `group` defaults to 4 for a resident corpus and to 1 when the clips are synthesised
    on the fly (the synthesis of the next chunk then runs under the engines' work on the current one; with groups it would
    have to run ahead of a whole group)."""
    if group is None:
        group = 4 if resident is not None else 1
    if0_chunk = max(1, int(chunk))
    chunk = if0_chunk * max(1, int(group))
    if compute is None:
        compute = _engine_compute
        engine_path = True
    else:
        engine_path = False
    lo, hi = partition(n_clips, world, rank)
    out = np.zeros((hi - lo, len(methods), 12), dtype=np.float64)
    spent = [0.0] * len(methods)
    global LAST_SYNTH_SECONDS
    LAST_SYNTH_SECONDS = 0.0
    # The next chunk is synthesised while the engines work on the current one: on the GPU the synthesis is a few hundred
    # memory-bound torch kernels on a stream of its own, enqueued before the (blocking) engine calls -- like a decoder
    # thread ahead of the compute in a corpus of files.  Only the wait that is left is booked as synthesis time.
    starts = list(range(lo, hi, chunk))
    on_gpu = synth_device is not None and str(synth_device).startswith("cuda")
    side_stream = None
    if on_gpu:
        import torch
        side_stream = torch.cuda.Stream(device=synth_device)

    def synth_ahead(c0):
        if resident is not None:
            return resident[c0][0], resident[c0][1], None
        ids = list(range(c0, min(c0 + chunk, hi)))
        if side_stream is None:
            return ids, synth_chunk(ids, fs, seconds, synth_device), None
        import torch
        with torch.cuda.stream(side_stream):
            x = synth_chunk(ids, fs, seconds, synth_device)
            ev = torch.cuda.Event()
            ev.record(side_stream)
        return ids, x, ev

    import threading

    def start_ahead(c0):   # host part (the partial tables: Python loops) and enqueueing in a thread of their own:
        box = []           # ctypes releases the GIL while the main thread is inside the engine

        def work():
            try:
                box.append(("ok", synth_ahead(c0)))
            except BaseException as exc:   # handed to the main thread, which re-raises it with its cause
                box.append(("err", exc))

        th = threading.Thread(target=work)
        th.start()
        return th, box

    ahead = start_ahead(starts[0]) if starts else None
    for ci, c0 in enumerate(starts):
        t_s = time.perf_counter()
        ahead[0].join()
        if not ahead[1] or ahead[1][0][0] != "ok":
            raise RuntimeError("corpus synthesis of the chunk at clip %d failed" % c0) from (ahead[1][0][1] if ahead[1] else None)
        ids, clips, ev = ahead[1][0][1]
        if ev is not None:
            ev.synchronize()   # the engines read the chunk through its raw pointer on their own streams
        ahead = start_ahead(starts[ci + 1]) if ci + 1 < len(starts) else None
        LAST_SYNTH_SECONDS += time.perf_counter() - t_s
        # synthesised on the GPU and consumed by the engine: the chunk stays in HBM (include/mpx.h, "where the samples
        # live"); a substituted compute function and a CPU synthesis get a host array
        clips = clips if (clips.is_cuda and engine_path) else clips.cpu().numpy()
        rows = slice(c0 - lo, c0 - lo + len(ids))

        failed = []

        def run(mi, m, fn):
            t0 = time.perf_counter()
            try:
                out[rows, mi] = fn()   # [n, L] array: packed as it is
            except BaseException as exc:   # re-raised by the caller's thread below: a dead side thread must not
                failed.append(exc)         # leave all-zero chroma behind a normal-looking summary
            spent[mi] += time.perf_counter() - t0

        side = None
        main_done = threading.Event()
        main_engine = None
        if overlap and engine_path and any(m not in (3, 4) for m in methods):
            from .engine import get_engine
            main_engine = get_engine(device)
        if overlap and engine_path and 3 in methods and len(methods) > 1:
            # Iterative-F0's front end is one serial chain per lane and leaves most issue slots of a SIMD free: it runs on
            # a second context and stream (ctypes releases the GIL) while the other methods go through the first one
            import threading
            mi3 = list(methods).index(3)
            eng2 = _second_engine(device)

            def if0_pieces():
                # the whole group in ONE engine call since round 4: above its workspace cap the library runs a clip list
                # in time slices with every clip in flight (include/mpx.h MPX_OPT_IF0_WORKSPACE_BYTES) -- until then
                # `if0_chunk` clips per call, each call a front-end launch of its own with one leftover-channel wave per clip
                return np.asarray(eng2.iterative_f0_batch(clips, fs, note_names=note_names))

            side = threading.Thread(target=run, args=(mi3, 3, if0_pieces))
            _start_side(side, main_engine, main_done)
        side4 = None
        if side is not None and 4 in methods and len(methods) > 2 and os.environ.get("MPX_CORPUS_CONTEXTS", "3") == "3":
            # ... and Prime-multiF0 (seven million small workgroups, latency-bound) on a third one
            mi4 = list(methods).index(4)
            eng3 = _second_engine((device, 3))
            side4 = threading.Thread(target=run, args=(mi4, 4, lambda: eng3.prime_multif0_batch(clips, fs,
                                                                                               note_names=note_names)))
            side4.start()   # (at once: its waves do not keep ESACF's workgroups out, and it is over before the front end's first round)
        for mi, m in enumerate(methods):
            if (side is not None and m == 3) or (side4 is not None and m == 4):
                continue
            if engine_path:
                run(mi, m, lambda m=m: compute(m, clips, fs, device, note_names))
            else:
                run(mi, m, lambda m=m: compute(m, clips, fs, device))
            if failed:
                break
        main_done.set()
        if side is not None:
            side.join()
        if side4 is not None:
            side4.join()
        if failed:
            if ahead is not None:   # do not leave the next chunk's synthesis running (and allocating) behind the exception
                ahead[0].join()
            raise failed[0]
    return lo, hi, out, spent


def gather_blocks(block, n_clips, world, rank, device=None, force=False):
    """The job's one collective: all_gather of the (padded) per-rank blocks -> [n_clips, M, 12] on every rank.  With one
    rank there is nothing to gather and the block is returned as it is, unless `force` (launch.wants_collective: a
    one-rank process group exists and the collective is to run over it all the same)."""
    if world == 1 and not force:
        return block
    import torch
    import torch.distributed as dist
    per = -(-int(n_clips) // world)  # ceil: blocks are padded to a common size
    pad = np.zeros((per,) + block.shape[1:], dtype=np.float64)
    pad[:block.shape[0]] = block
    t = torch.from_numpy(pad)
    if device is not None:
        t = t.to(device)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    rows = []
    for r, p in enumerate(parts):
        lo, hi = partition(n_clips, world, r)
        rows.append(p[:hi - lo].cpu().numpy())
    return np.concatenate(rows, axis=0)


def summarise(chroma, methods, seconds_per_method, n_clips, wall):
    res = {"clips": int(n_clips), "wall_seconds": wall, "clips_per_s": n_clips / wall if wall > 0 else None,
           "methods": {}}
    for mi, m in enumerate(methods):
        keys = {}
        for row in chroma[:, mi]:
            k = Chromagram(row).key()
            keys[k] = keys.get(k, 0) + 1
        res["methods"][str(m)] = {
            "name": METHODS[m].display_name(),
            "seconds_rank_max": seconds_per_method[mi],
            "mean_chroma": [float(v) for v in chroma[:, mi].mean(axis=0)],
            "first_clip": repr(Chromagram(chroma[0, mi])) if len(chroma) else None,
            "keys": dict(sorted(keys.items(), key=lambda kv: -kv[1])[:6]),
        }
    return res


def main(argv=None, compute=None, device="cuda", backend="nccl"):
    """`compute`, `device`, `backend`: injection points for the CPU tests (a checker instead of the engine, CPU tensors,
    gloo): the command line never sets them -- without a GPU and the HIP library this driver fails, it has no fallback."""
    import argparse
    import os
    ap = argparse.ArgumentParser(description="all four methods over a synthetic corpus, clip-sharded over the visible GPUs")
    ap.add_argument("--clips", type=int, default=4096)
    ap.add_argument("--methods", default="1,2,3,4")
    ap.add_argument("--fs", type=int, default=22050)
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--chunk", type=int, default=1024, help="clips per Iterative-F0 engine call")
    ap.add_argument("--group", type=int, default=None, help="chunks the other methods take per engine call (default: 1; 4 for a resident corpus)")
    ap.add_argument("--out", default=None, help="write per-clip chroma [clips, methods, 12] to this .npz")
    ap.add_argument("--no-overlap", action="store_true", help="run Iterative-F0 after the other methods instead of next to them")
    ap.add_argument("--note-names", choices=("unicode", "ascii"), default="unicode",
                    help="spelling of sharps by the librosa the reference runs with (include/mpx.h MPX_NOTES_*)")
    ap.add_argument("--force-collective", action="store_true",
                    help="create the process group and run the closing all_gather even with ONE rank (RCCL on a one-GPU box)")
    args = ap.parse_args(argv)
    methods = [int(m) for m in args.methods.split(",")]
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
    t0 = time.perf_counter()
    lo, hi, block, spent = run_corpus(args.clips, methods, args.fs, args.seconds, args.chunk, rank, world, local,
                                      compute=compute, synth_device=dev if on_gpu else None, overlap=not args.no_overlap,
                                      note_names=args.note_names, group=args.group)
    chroma = gather_blocks(block, args.clips, world, rank, dev if (use_dist and on_gpu) else None, force=use_dist)
    wall = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([wall] + spent, dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall, spent = float(tt[0]), [float(v) for v in tt[1:]]
    if rank == 0:
        res = summarise(chroma, methods, spent, args.clips, wall)
        res["n_gpus"] = world
        res["collective"] = ("%s all_gather over %d rank(s)" % (backend, world)) if use_dist else None
        res["synthesis_seconds_rank0"] = LAST_SYNTH_SECONDS
        if args.out:
            np.savez_compressed(args.out, chroma=chroma, methods=np.array(methods))
        print(json.dumps(res))
    if use_dist:
        dist.destroy_process_group()
    return 0
