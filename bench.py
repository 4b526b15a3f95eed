#!/usr/bin/env python3
"""Headline benchmark: frames/s for STFT -> harmonic-energy chromagram
(4096-pt FFT, hop 1024, 44.1 kHz, 8192-frame batches) -- BASELINE.json configs[1].

  python bench.py --gpus N --steps K --warmup W
For N > 1 the driver launches one rank per GPU with torch.distributed.run; the
frames shard across ranks (each rank owns its own 8192-frame batch, no
data-path collective) and the per-step 12-vectors are gathered once at the end
with RCCL (backend "nccl").  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FS, N_FFT, HOP, FRAMES = 44100, 4096, 1024, 8192
F_ALG = int(2.5 * 4096 * 12 + 4096)  # 2.5 N log2 N + N at N = 4096
B_ALG = 4 * HOP + 48          # compulsory HBM bytes per frame, overlapped-signal input
PREHEAT_MS = 100              # untimed launches before the W warm-up steps: clock ramp of a cold device (see main)
HBM_PEAK = 8.0e12             # MI355X_MICROARCH.md: 8.0 TB/s spec


def synth_signal(seed, frames=FRAMES):
    """SURVEY.md 8(d) M-HE: decaying-harmonic notes at random MIDI 36-84, 0.5 s each,
    + white noise at -40 dBFS, peak 0.9.  float32, (frames-1)*hop + N samples."""
    import numpy as np
    n = (frames - 1) * HOP + N_FFT
    rng = np.random.default_rng(seed)
    seg = FS // 2
    x = np.zeros(n, dtype=np.float64)
    t = np.arange(seg) / FS
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


def cpu_baseline(x, budget_s=12.0):
    """The NumPy oracle (a port of the reference's per-frame math, vectorised over
    frames) on one host core, on a bounded sample of the same workload."""
    import numpy as np
    from oracle import harmonic_energy as o_he
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=1)
    except Exception:
        limiter = None
    chunk = 1024
    done, t0 = 0, time.perf_counter()
    f = 0
    while True:
        lo = (f % (FRAMES // chunk)) * chunk * HOP
        o_he.he_frames(x[lo:lo + (chunk - 1) * HOP + N_FFT], FS, N_FFT, HOP)
        done += chunk
        f += 1
        el = time.perf_counter() - t0
        if el > budget_s:
            break
    # the reference's own loop structure (one frame at a time), for honesty
    t1 = time.perf_counter()
    k = 256
    for i in range(k):
        o_he.he_frames(x[i * HOP:i * HOP + N_FFT], FS, N_FFT)
    loop_rate = k / (time.perf_counter() - t1)
    if limiter is not None:
        limiter.unregister() if hasattr(limiter, "unregister") else None
    return {"value": done / el, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d frames (N=4096, hop=1024) of the bench signal through oracle/harmonic_energy.py "
                      "(numpy.fft.rfft, float64), 1 thread, %.1f s" % (done, el),
            "per_frame_loop_frames_per_s": loop_rate, "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 0.15 s of GPU work.  The device needs on the order of 0.05 s under load to reach its sustained clock; a
    # 20 + 200-step run (15 ms) times the ramp and reports the kernel 10 % slower than it runs a moment later.
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--streams", type=int, default=2,
                    help="batches in flight: step i goes to context/stream i %% S (1 = strictly one launch after the other)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--f32", action="store_true", help="opt-in fp32 engine (not the headline)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)

    import numpy as np
    import torch  # before the HIP library: one shared libamdhip64 in the process
    import torch.distributed as dist
    import chord_detection_amd as cd

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    # Two batches in flight: consecutive steps alternate between two contexts (own stream, own reduction scratch), so the
    # ramp of one launch -- dispatch, table loads, the first un-prefetched frame -- and its tail -- the last workgroups,
    # the in-kernel reduction -- run while the other launch has the machine.  A step is still one launch over one
    # 8192-frame batch; measured 55.4 us/step with one stream, 42.6 with two, no further gain with three.
    nstreams = max(1, args.streams)
    engs = [cd.Engine(local_rank, f32=args.f32) for _ in range(nstreams)]
    eng = engs[0]
    x_host = synth_signal(20260101 + rank)
    x = torch.from_numpy(x_host).to(dev)
    n = x.numel()
    steps, warmup = args.steps, args.warmup
    d_frames = torch.empty((FRAMES, 12), dtype=torch.float64, device=dev)
    d_sums = torch.zeros((max(steps, warmup, nstreams, 1), 12), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()

    def step(i):
        # the product path for one signal: chroma summed over frames, no per-frame rows
        engs[i % nstreams].harmonic_energy_dev(x.data_ptr(), n, FS, N_FFT, HOP, None, d_sums.data_ptr() + i * 96)

    def sync_engines():
        for e in engs:
            e.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Untimed pre-heat: the same launch for PREHEAT_MS of wall time, so that the device is at its sustained clock
    # whatever W and K are (measured: 62.3 us/step with W=20, K=200 from a cold device, 55.4 at any larger K).
    t_pre = time.perf_counter()
    while 1e3 * (time.perf_counter() - t_pre) < PREHEAT_MS:
        for j in range(64):
            step(j % max(nstreams, 1))
        sync_engines()
    for i in range(warmup):
        step(i)
    if world > 1:  # the job's one collective, once untimed: RCCL sets its rings up on first use
        sync_engines()
        dist.all_gather([torch.empty_like(d_sums) for _ in range(world)], d_sums)
    barrier()
    t0 = time.perf_counter()
    th0 = time.perf_counter()
    for i in range(steps):
        step(i)
    host_enqueue_ms = 1e3 * (time.perf_counter() - th0) / max(steps, 1)
    gathered = None
    if world > 1:
        sync_engines()
        gathered = [torch.empty_like(d_sums) for _ in range(world)]
        dist.all_gather(gathered, d_sums)          # one RCCL gather of the 12-vectors, at the end
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # one launch after the other on ONE stream, HIP events on that stream: the full step (with the in-kernel reduction) ...
    sync_engines()
    reps = max(min(steps, 2000), 50)
    d_seq = torch.zeros(12, dtype=torch.float64, device=dev)
    eng.timer_begin()
    for _ in range(reps):
        eng.harmonic_energy_dev(x.data_ptr(), n, FS, N_FFT, HOP, None, d_seq.data_ptr())
    step_ms_events = eng.timer_end() / reps
    # ... and the dominant kernel alone (per-frame rows out, no final 12-vector reduction): the roofline's launch duration
    eng.synchronize()
    reps = max(steps, 50)
    eng.timer_begin()
    for _ in range(reps):
        eng.harmonic_energy_dev(x.data_ptr(), n, FS, N_FFT, HOP, d_frames.data_ptr(), None)
    kern_ms = eng.timer_end() / reps
    achieved = B_ALG * FRAMES / (kern_ms * 1e-3)

    # sanity: the benchmarked output is the real thing (checked against the oracle on a few frames)
    from oracle import harmonic_energy as o_he
    got = d_frames[:4].cpu().numpy()
    want = o_he.he_frames(x_host[:3 * HOP + N_FFT], FS, N_FFT, HOP)
    tol = 2e-4 if args.f32 else 1e-9
    if not np.allclose(got, want, rtol=tol):
        sys.exit("bench: GPU output does not match the oracle")
    sums = d_sums[:steps].cpu().numpy()
    if steps and not np.allclose(sums, sums[0], rtol=0, atol=0):
        sys.exit("bench: per-step results differ (non-deterministic)")
    if steps and not np.allclose(sums[0], d_frames.sum(0).cpu().numpy(), rtol=1e-10):
        sys.exit("bench: fused chroma sum does not match the sum of the per-frame rows")

    traffic = None
    try:  # HBM bytes per launch from the last PMC probe of this kernel (bench.py cannot run rocprofv3 on itself)
        with open(os.path.join(ROOT, "profiles", "traffic_latest.json")) as fh:
            tr = json.load(fh)
        if tr.get("kernel") == "he_kernel<4096,256,%s>" % ("float" if args.f32 else "double"):
            traffic = tr["bytes_per_launch"]
    except Exception:
        traffic = None
    if rank == 0:
        total_frames = FRAMES * world * steps
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
            "config": {"workload": "Harmonic Energy STFT->chromagram, 8192 synthetic 44.1 kHz frames per GPU, "
                                   "4096-pt FFT hop 1024 (BASELINE.json configs[1])",
                       "frames_per_gpu": FRAMES, "fft": N_FFT, "hop": HOP, "fs": FS, "untimed_preheat_ms": PREHEAT_MS,
                       "batches_in_flight": nstreams,
                       "sharding": "frames per rank, no data-path collective; one RCCL all_gather of 12-vectors at the end"},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": traffic,
                         "traffic_note": "bytes/launch, rocprofv3 FETCH_SIZE(x1.99 calibrated)+WRITE_SIZE, profiles/r1/traffic_v5.txt; algorithmic = %d" % (B_ALG * FRAMES),
                         "kernel": "he_kernel<4096,256,%s>" % ("float" if args.f32 else "double"),
                         "kernel_ms": kern_ms, "bytes_per_frame": B_ALG, "frames_per_launch": FRAMES,
                         "step_ms_hip_events": step_ms_events, "host_enqueue_ms_per_step": host_enqueue_ms,
                         # achieved / frac above are per launch: the kernel's own duration, one launch at a time.  With
                         # `batches_in_flight` launches overlapping, the device moves the algorithmic bytes of all
                         # timed launches in the timed region at this rate (per GPU):
                         "in_flight": {"batches": nstreams,
                                       "achieved": B_ALG * FRAMES * steps / max(elapsed, 1e-12) / 1e9, "unit": "GB/s",
                                       "frac": B_ALG * FRAMES * steps / max(elapsed, 1e-12) / HBM_PEAK},
                         # the binding roof of an fp64 LDS FFT is the vector ALU, not HBM: standard
                         # real-FFT count 2.5 N log2 N + N window multiplies per frame against the fp64 vector
                         # peak (half the 157.3 TFLOP/s FP32 vector rate of MI355X_MICROARCH.md)
                         "secondary": {"bound": "valu_f32" if args.f32 else "valu_f64",
                                       "achieved": FRAMES * F_ALG / (kern_ms * 1e-3) / 1e12,
                                       "peak": 157.3 if args.f32 else 78.65, "unit": "TFLOP/s",
                                       "frac": FRAMES * F_ALG / (kern_ms * 1e-3) / 1e12 / (157.3 if args.f32 else 78.65),
                                       "flops_per_frame": F_ALG}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(x_host)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
