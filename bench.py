#!/usr/bin/env python3
"""Headline benchmark: frames/s for STFT -> harmonic-energy chromagram
(4096-pt FFT, hop 1024, 44.1 kHz, 8192-frame batches) -- BASELINE.json configs[1] -- plus, in the same JSON
line under "workloads", the other BASELINE workloads and the north star's Target:

  esacf_clips_4096         configs[2]  ESACF over 4096 polyphonic 2 s clips @44.1 kHz (46.4 ms frames, 2046 samples)
  esacf_stft_8192          Target      STFT -> ESACF -> chromagram, ONE signal, 8192 frames, N=4096 hop 1024
  corpus_4096_all_methods  configs[3]  all four methods over 4096 clips per GPU (2 s @22.05 kHz), one gather
  if0_stream_1h            configs[4]  Iterative-F0 over a 1 h stream @44.1 kHz, time-sharded over the GPUs
  he_default_8192          reference default shape: Harmonic Energy, 8192-sample frames, hop = frame, 22.05 kHz clips

  python bench.py --gpus N --steps K --warmup W
For N > 1 the driver launches one rank per GPU with torch.distributed.run; the frames shard across ranks (each
rank owns its own 8192-frame batches, no data-path collective) and the per-step 12-vectors are gathered once at
the end with RCCL (backend "nccl").  Rank 0 prints one JSON line.  `--force-collective` (or a one-rank launch through
torch.distributed.run) creates the communicator and runs that gather with ONE rank as well.

The CPU legs (`cpu_baseline`: the NumPy oracle on one core and on all physical cores of the host) run FIRST, before
anything touches the GPU: they fork worker processes, and a process that has initialised HIP must not be forked
around lightly.

This file is the timed loop of the headline and the record's assembly; the parts live in benchlib/ (config: sizes and
peaks, synth: inputs, cpu_legs: oracle timings, roofline: work models, record: the compact line, workloads: the
secondary configs)."""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib import config as K                                                            # noqa: E402
from benchlib.config import FS, N_FFT, HOP, F_ALG, B_ALG, HBM_PEAK, F64_PEAK, CFG            # noqa: E402,F401
from benchlib.synth import synth_signal, synth_clips_numpy, synth_signal_device              # noqa: E402,F401
from benchlib.cpu_legs import cpu_baselines, host_cpu_info, _cpu_worker, _CPU_INPUT          # noqa: E402,F401
from benchlib.roofline import measured_traffic, he_kernel_name, traffic_from                               # noqa: E402
from benchlib.record import compact_record, write_full_record, COMPACT_LIMIT                 # noqa: E402,F401
from benchlib.workloads import WORKLOADS                                                     # noqa: E402


def __getattr__(name):   # bench.FRAMES / bench.NSIG / bench.PREHEAT_MS: the live values of benchlib.config
    if name in ("FRAMES", "NSIG", "PREHEAT_MS"):
        return getattr(K, name)
    raise AttributeError(name)


def main():
    # tests only: a stand-in for torch.cuda + the HIP engine, so that the world > 1 code below runs over gloo on CPU
    stub = importlib.import_module(os.environ["MPX_BENCH_STUB"]) if os.environ.get("MPX_BENCH_STUB") else None
    if stub is not None:
        stub.configure(K)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 0.1 s of GPU work.  The device needs on the order of 0.05 s under load to reach its sustained clock; the
    # untimed pre-heat below makes the figure independent of K and W.
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--streams", type=int, default=4,
                    help="batches in flight: step i goes to context/stream i %% S (1 = strictly one launch after the other)")
    ap.add_argument("--repeats", type=int, default=0,
                    help="the K-step timed loop is run this many times, each bracketed by barrier + synchronize (EXACTLY K steps "
                         "per timed region); the MEDIAN is reported.  0 (default): 5 repeats, 25 when K < 200 (the driver's short "
                         "run, K = 20, is 0.8 ms of GPU time per repeat: five samples of it are a thin median)")
    ap.add_argument("--force-collective", action="store_true",
                    help="create the process group and run the job's all_gather even with ONE rank (RCCL comm init + collective "
                         "on a one-GPU box); implied by a launch through torch.distributed.run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="skip the secondary workloads")
    ap.add_argument("--workloads", default="esacf_clips_4096,esacf_stft_8192,corpus_4096_all_methods,if0_stream_1h,he_default_8192")
    ap.add_argument("--signals", type=int, default=K.NSIG, help="distinct input signals the steps rotate over")
    ap.add_argument("--f32", action="store_true", help="opt-in fp32 engine (not the headline)")
    ap.add_argument("--full-json", default="bench_full.json",
                    help="file that receives the full record; stdout's last line is the compact one (<= 6 KB)")
    args = ap.parse_args()

    # Nothing in the environment may change what is measured: the release library reads no development switches, and this
    # script refuses to run with any MPX_* variable other than its own two set (MPX_LIB_PATH would swap the library,
    # MPX_DETERMINISTIC the fit scheduling).
    foreign = sorted(k for k in os.environ if k.startswith("MPX_") and k not in ("MPX_BENCH_STUB", "MPX_BENCH_CPU_BUDGET", "MPX_BENCH_DUMP_GATHER"))
    if foreign:
        sys.exit("bench.py: unset %s (bench numbers are taken with the default library and defaults only)" % ", ".join(foreign))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    launched = "TORCHELASTIC_RUN_ID" in os.environ   # a torch.distributed.run worker, possibly the only one
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # rank 0 at N=1 only, and before the GPU is touched
        cpu = cpu_baselines(budget_s=float(os.environ.get("MPX_BENCH_CPU_BUDGET", "6")))

    import numpy as np
    import torch  # before the HIP library: one shared libamdhip64 in the process
    import torch.distributed as dist
    import chord_detection_amd as cd
    from chord_detection_amd import launch

    # One rank has nothing to gather; with --force-collective (or under the launcher) the communicator is created and the
    # job's collectives run over the one rank all the same: the RCCL path of a multi-GPU launch on a one-GPU box.
    use_dist = world > 1 or args.force_collective or launched
    if stub is None:
        dev = torch.device("cuda", local_rank)
        backend = "nccl"
        if use_dist:   # the FIRST thing this process does on the GPU: the communicator is bound to the device here
            launch.init_group(backend, dev)
        from chord_detection_amd import _lib
        if _lib.load().mpx_dev_knobs():
            sys.exit("bench.py: the loaded library was built with -DMPX_DEV_KNOBS (development switches); use the release build")
        torch.cuda.set_device(local_rank)
        make_engine = lambda: cd.Engine(local_rank, f32=args.f32)
    else:
        dev, make_engine, backend = torch.device("cpu"), stub.Engine, "gloo"
        if use_dist:
            launch.init_group(backend, None)

    def dev_sync():
        if dev.type == "cuda":
            torch.cuda.synchronize()

    def barrier():
        if use_dist:
            dist.barrier()
        dev_sync()

    # Batches in flight: consecutive steps go round-robin to `--streams` contexts (own stream, own reduction scratch), so the
    # ramp of one launch -- dispatch, table loads, the first un-prefetched frame -- and its tail -- the last workgroups,
    # the small reduction launch -- run while other launches have the machine.  he_wave_kernel takes a CU's whole LDS, so
    # launches do not share CUs, they fill each other's edges: 43.4 us per step one at a time, 39.4 with two in flight,
    # 38.4 with three, 37.0 with four, 38.1 / 37.1 with six / eight (MI355X).  A step is still one launch over one
    # 8192-frame batch.  Step i reads signal i % NSIG: together the signals exceed the 256 MiB Infinity Cache, so a
    # step's input comes from HBM, not from a cache warmed by the step before.
    nstreams = max(1, args.streams)
    nsig = max(1, args.signals)
    engs = [make_engine() for _ in range(nstreams)]
    eng = engs[0]
    sigs = [synth_signal_device(20260101 + 1000 * rank + k, dev, K.FRAMES) for k in range(nsig)]
    x_host = sigs[0].cpu().numpy()
    n = sigs[0].numel()
    steps, warmup = args.steps, args.warmup
    d_frames = torch.empty((K.FRAMES, 12), dtype=torch.float64, device=dev)
    d_sums = torch.zeros((max(steps, warmup, nstreams, nsig, 1), 12), dtype=torch.float64, device=dev)
    dev_sync()

    def step(i, e=None):
        # the product path for one signal: chroma summed over frames, no per-frame rows
        (e or engs[i % nstreams]).harmonic_energy_dev(sigs[i % nsig].data_ptr(), n, K.FS, K.N_FFT, K.HOP, None,
                                                     d_sums.data_ptr() + i * 96)

    def sync_engines():
        for e in engs:
            e.synchronize()

    # Untimed pre-heat: the same launches for PREHEAT_MS of wall time, so that the device is at its sustained clock
    # whatever W and K are (measured: 62.3 us/step with W=20, K=200 from a cold device, 55.4 at any larger K).
    t_pre = time.perf_counter()
    while 1e3 * (time.perf_counter() - t_pre) < K.PREHEAT_MS:
        for j in range(64):
            step(j % max(nstreams, nsig))
        sync_engines()
    for i in range(warmup):
        step(i)
    if use_dist:  # the job's one collective, once untimed: RCCL sets its rings up on first use
        sync_engines()
        dist.all_gather([torch.empty_like(d_sums) for _ in range(world)], d_sums)
    # The timed region, `repeats` times: EXACTLY K steps between barrier + synchronize on both sides, the rank maximum of
    # each repeat, and the median of the repeats is what `value` is computed from (every repeat is listed in the output).
    repeats = args.repeats if args.repeats > 0 else (25 if steps < 200 else 5)
    rep_s, rep_steps_s, rep_gather_s, host_enqueue_ms, gathered = [], [], [], 0.0, None
    gather_out = torch.empty((world,) + tuple(d_sums.shape), dtype=d_sums.dtype, device=dev) if use_dist and backend == "nccl" else None
    for rep in range(repeats):
        barrier()
        t0 = time.perf_counter()
        th0 = time.perf_counter()
        for i in range(steps):
            step(i)
        host_enqueue_ms = 1e3 * (time.perf_counter() - th0) / max(steps, 1)
        el_steps = None
        if use_dist:
            sync_engines()
            el_steps = time.perf_counter() - t0        # this rank's K steps alone: the clock read before the job's one gather
            if gather_out is not None:                 # RCCL: into one preallocated tensor (no list of outputs to build per repeat)
                dist.all_gather_into_tensor(gather_out, d_sums)
                gathered = gather_out
            else:
                gathered = [torch.empty_like(d_sums) for _ in range(world)]
                dist.all_gather(gathered, d_sums)      # one gather of the 12-vectors, at the end
        else:
            sync_engines()
        # the clock stops when THIS rank's K steps (and its part of the gather) are done; the closing barrier + synchronize
        # follow, and the maximum over the ranks is when the job was done -- the barrier's own latency (tens of microseconds
        # of a 0.8 ms region at the driver's K = 20) is not part of the steps
        dev_sync()
        el = time.perf_counter() - t0
        barrier()
        # [whole region, the steps alone, the gather alone]: the rank maximum of each, so that a weak-scaling loss of a short
        # region (0.8 ms at the driver's K = 20) can be split into "the steps got slower" and "the gather's latency"
        t = torch.tensor([el, el if el_steps is None else el_steps, 0.0 if el_steps is None else el - el_steps],
                         dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rep_s.append(float(t[0].item()))
        rep_steps_s.append(float(t[1].item()))
        rep_gather_s.append(float(t[2].item()))
    if rank == 0 and gathered is not None and os.environ.get("MPX_BENCH_DUMP_GATHER"):   # tests: what the gather delivered
        g = gathered if torch.is_tensor(gathered) else torch.stack(list(gathered))
        np.save(os.environ["MPX_BENCH_DUMP_GATHER"], g.cpu().numpy())
    elapsed = sorted(rep_s)[len(rep_s) // 2]
    elapsed_steps = sorted(rep_steps_s)[len(rep_steps_s) // 2]
    gather_s = sorted(rep_gather_s)[len(rep_gather_s) // 2]
    sums = d_sums[:steps].cpu().numpy()

    # the same K steps strictly one launch after the other on ONE context/stream (what the per-launch roofline describes)
    one_s = []
    for rep in range(repeats):
        barrier()
        t1 = time.perf_counter()
        for i in range(steps):
            step(i, eng)
        eng.synchronize()
        el1 = time.perf_counter() - t1
        barrier()
        one = torch.tensor([el1], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(one, op=dist.ReduceOp.MAX)
        one_s.append(float(one.item()))
    elapsed_one = sorted(one_s)[len(one_s) // 2]

    # HIP events on that stream: the full step (with the in-kernel reduction) ...
    reps = max(min(steps, 2000), 50)
    d_seq = torch.zeros(12, dtype=torch.float64, device=dev)
    eng.timer_begin()
    for r in range(reps):
        eng.harmonic_energy_dev(sigs[r % nsig].data_ptr(), n, K.FS, K.N_FFT, K.HOP, None, d_seq.data_ptr())
    step_ms_events = eng.timer_end() / reps
    # ... and the dominant kernel alone (per-frame rows out, no final 12-vector reduction): the roofline's launch duration
    eng.synchronize()
    reps = max(steps, 50)
    eng.timer_begin()
    for r in range(reps):
        eng.harmonic_energy_dev(sigs[r % nsig].data_ptr(), n, K.FS, K.N_FFT, K.HOP, d_frames.data_ptr(), None)
    kern_ms = eng.timer_end() / reps
    achieved = K.B_ALG * K.FRAMES / (kern_ms * 1e-3)

    # sanity: the benchmarked output is the real thing (checked against the oracle on a few frames)
    from oracle import harmonic_energy as o_he
    eng.harmonic_energy_dev(sigs[0].data_ptr(), n, K.FS, K.N_FFT, K.HOP, d_frames.data_ptr(), None)
    eng.synchronize()
    got = d_frames[:4].cpu().numpy()
    want = o_he.he_frames(x_host[:3 * K.HOP + K.N_FFT], K.FS, K.N_FFT, K.HOP)
    tol = 2e-4 if args.f32 else 1e-9
    if not np.allclose(got, want, rtol=tol):
        sys.exit("bench: GPU output does not match the oracle")
    if steps > nsig and not np.array_equal(sums[:steps - nsig], sums[nsig:steps]):
        sys.exit("bench: results of the same signal differ between steps (non-deterministic)")
    if steps and not np.allclose(sums[0], d_frames.sum(0).cpu().numpy(), rtol=1e-10):
        sys.exit("bench: fused chroma sum does not match the sum of the per-frame rows")

    # HBM bytes per launch from the last PMC collection of this kernel (bench.py cannot run rocprofv3 on itself)
    traffic, traffic_note = measured_traffic("he", "he_kernel")

    workloads = {}
    if not args.headline_only:
        want_w = [w for w in args.workloads.split(",") if w]
        ctxw = dict(cd=cd, torch=torch, dist=dist, np=np, dev=dev, rank=rank, world=world, local_rank=local_rank,
                    eng=eng, engs=engs, make_engine=make_engine, sigs=sigs, x_host=x_host, barrier=barrier,
                    dev_sync=dev_sync, stub=stub, cpu=cpu, use_dist=use_dist)
        for name in want_w:
            barrier()
            rec = WORKLOADS[name](ctxw)
            if rec is not None:
                workloads[name] = rec

    if rank == 0:
        total_frames = K.FRAMES * world * steps
        kname = he_kernel_name(args.f32)
        out = {
            "metric": "frames/sec STFT->chromagram (4096-pt FFT, hop 1024)",
            "value": total_frames / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": 1e3 * elapsed / max(steps, 1),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.f32 else "f64",
            "data": "synthetic",
            "engine": "hip" if stub is None else "stub",
            "ms_per_step_repeats": [1e3 * v / max(steps, 1) for v in rep_s],
            # the same timed regions with the clock read BEFORE the job's one all_gather (rank maximum, median of the repeats),
            # and the gather + its synchronize on their own: value = frames / (steps + gather); value_steps_only = frames / steps
            "value_steps_only": total_frames / max(elapsed_steps, 1e-12),
            "ms_per_step_steps_only": 1e3 * elapsed_steps / max(steps, 1),
            "gather_ms": 1e3 * gather_s if use_dist else None,
            "gather_ms_repeats": [1e3 * v for v in rep_gather_s] if use_dist else None,
            "value_one_in_flight": total_frames / elapsed_one,
            "ms_per_step_one_in_flight": 1e3 * elapsed_one / max(steps, 1),
            "config": {"workload": "Harmonic Energy STFT->chromagram, 8192 synthetic 44.1 kHz frames per GPU, "
                                   "4096-pt FFT hop 1024 (BASELINE.json configs[1])",
                       "frames_per_gpu": K.FRAMES, "fft": K.N_FFT, "hop": K.HOP, "fs": K.FS, "untimed_preheat_ms": K.PREHEAT_MS,
                       "repeats": repeats, "repeat_statistic": "median",
                       "batches_in_flight": nstreams, "distinct_input_signals": nsig,
                       "input_bytes_rotated_over": int(nsig * n * 4),
                       "sharding": "frames per rank, no data-path collective; one RCCL all_gather of 12-vectors at the end",
                       "collective": ("%s all_gather of [steps, 12] inside every timed repeat, %d rank(s)" % (backend, world))
                       if use_dist else None},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": K.HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / K.HBM_PEAK, "traffic": traffic,
                         # where `traffic` was read: bench.py cannot run rocprofv3 on itself, so it is the committed PMC
                         # collection of the same launch shape (profiles/traffic_latest.json), NOT a counter of this run
                         "traffic_from": traffic_from("he"),
                         "traffic_note": "bytes/launch; %s; algorithmic = %d" % (traffic_note, K.B_ALG * K.FRAMES),
                         "compulsory_bytes": K.B_ALG * K.FRAMES,
                         "wasted_traffic_ratio": (traffic / (K.B_ALG * K.FRAMES)) if traffic else None,
                         "kernel": kname,
                         "kernel_ms": kern_ms, "bytes_per_frame": K.B_ALG, "frames_per_launch": K.FRAMES,
                         "step_ms_hip_events": step_ms_events, "host_enqueue_ms_per_step": host_enqueue_ms,
                         # achieved / frac above are per launch: the kernel's own duration, one launch at a time.  With
                         # `batches_in_flight` launches overlapping, the device moves the algorithmic bytes of all
                         # timed launches in the timed region at this rate (per GPU):
                         "in_flight": {"batches": nstreams,
                                       "achieved": K.B_ALG * K.FRAMES * steps / max(elapsed, 1e-12) / 1e9, "unit": "GB/s",
                                       "frac": K.B_ALG * K.FRAMES * steps / max(elapsed, 1e-12) / K.HBM_PEAK},
                         # the binding roof of an fp64 LDS FFT is the vector ALU, not HBM: standard
                         # real-FFT count 2.5 N log2 N + N window multiplies per frame against the fp64 vector
                         # peak (half the 157.3 TFLOP/s FP32 vector rate of MI355X_MICROARCH.md)
                         "secondary": {"bound": "valu_f32" if args.f32 else "valu_f64",
                                       "achieved": K.FRAMES * K.F_ALG / (kern_ms * 1e-3) / 1e12,
                                       "peak": 157.3 if args.f32 else 78.65, "unit": "TFLOP/s",
                                       "frac": K.FRAMES * K.F_ALG / (kern_ms * 1e-3) / 1e12 / (157.3 if args.f32 else 78.65),
                                       "flops_per_frame": K.F_ALG}},
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu["he"]
        if workloads:
            out["workloads"] = workloads
        # The whole record (every kernel's roofline, every CPU leg: ~25 KB) goes to a file; stdout's LAST line is the
        # compact record the driver parses (round 3's single 24 KB line outgrew the 8 KB the driver keeps).
        full_path = write_full_record(out, args.full_json)
        line = json.dumps(compact_record(out, full_path), separators=(",", ":"))
        if len(line) > COMPACT_LIMIT:
            sys.exit("bench.py: compact record is %d bytes (limit %d)" % (len(line), COMPACT_LIMIT))
        print(line, flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
