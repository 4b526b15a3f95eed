#!/usr/bin/env python3
"""Headline benchmark: frames/s for STFT -> harmonic-energy chromagram
(4096-pt FFT, hop 1024, 44.1 kHz, 8192-frame batches) -- BASELINE.json configs[1] -- plus, in the same JSON
line under "workloads", the other BASELINE workloads and the north star's Target:

  esacf_clips_4096         configs[2]  ESACF over 4096 polyphonic 2 s clips @44.1 kHz (46.4 ms frames, 2046 samples)
  esacf_stft_8192          Target      STFT -> ESACF -> chromagram, ONE signal, 8192 frames, N=4096 hop 1024
  corpus_4096_all_methods  configs[3]  all four methods over 4096 clips per GPU (2 s @22.05 kHz), one gather
  if0_stream_1h            configs[4]  Iterative-F0 over a 1 h stream @44.1 kHz, time-sharded over the GPUs

  python bench.py --gpus N --steps K --warmup W
For N > 1 the driver launches one rank per GPU with torch.distributed.run; the frames shard across ranks (each
rank owns its own 8192-frame batches, no data-path collective) and the per-step 12-vectors are gathered once at
the end with RCCL (backend "nccl").  Rank 0 prints one JSON line.

The CPU legs (`cpu_baseline`: the NumPy oracle on one core and on all physical cores of the host) run FIRST, before
anything touches the GPU: they fork worker processes, and a process that has initialised HIP must not be forked
around lightly.
"""
import argparse
import importlib
import json
import math
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
F64_PEAK = 78.65e12           # fp64 vector: half the 157.3 TFLOP/s FP32 vector rate (same guide)
NSIG = 9                      # distinct input signals the steps rotate over: 9 x 33.5 MB = 302 MB > the 256 MiB MALL
# sizes of the secondary workloads (a stand-in backend for the CPU tests shrinks them)
CFG = {"esacf_clips": 4096, "esacf_fs": 44100, "esacf_clip_seconds": 2.0, "corpus_clips_per_gpu": 4096,
       "corpus_fs": 22050, "stream_seconds": 3600.0, "stream_fs": 44100, "if0_frame": 8192}


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


def synth_signal_device(seed, dev, frames=FRAMES):
    """The same recipe as synth_signal, evaluated with torch on `dev` (a few tensor ops instead of ~9000 NumPy ones per
    signal: the bench rotates over NSIG of them).  Not sample-identical to synth_signal (other random streams)."""
    import numpy as np
    import torch
    n = (frames - 1) * HOP + N_FFT
    rng = np.random.default_rng(seed)
    seg = FS // 2
    nseg = -(-n // seg)
    tab = np.zeros((nseg, 24, 3))      # [segment, 6 notes x 4 harmonics, (angular frequency, phase, amplitude)]
    for s in range(nseg):
        for k in range(int(rng.integers(3, 7))):
            f0 = 440.0 * 2.0 ** ((int(rng.integers(36, 85)) - 69) / 12.0)
            ph = rng.uniform(0, 2 * np.pi)
            for h in range(1, 5):
                tab[s, 4 * k + h - 1] = (2 * np.pi * f0 * h, ph, 0.5 ** (h - 1))
    t = torch.arange(seg, dtype=torch.float64, device=dev) / FS
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


# ---------------------------------------------------------------------------------------------------------------
# CPU legs: the oracle (a NumPy port of the reference's math) on the host cores of this box.  Reported, not a target.
# ---------------------------------------------------------------------------------------------------------------
def host_cpu_info():
    model, cores = None, set()
    try:
        phys = core = None
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name") and model is None:
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                    phys = core = None
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    physical = min(len(cores), usable) if cores else usable
    quota = None   # a container's CPU-time limit in cores (cgroup v2 cpu.max / v1 cfs quota), if any
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda v: None if v[0] == "max" else float(v[0]) / float(v[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", None)):
        try:
            with open(path) as fh:
                v = fh.read().split()
            if parse is None:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                    per = float(fh.read().split()[0])
                quota = None if float(v[0]) <= 0 else float(v[0]) / per
            else:
                quota = parse(v)
            break
        except (OSError, ValueError, IndexError):
            continue
    workers = max(1, physical)
    if quota is not None:
        workers = max(1, min(workers, int(quota)))
    return {"model": model, "logical": os.cpu_count(), "usable": usable, "physical": max(1, physical),
            "cgroup_cpu_quota": quota, "workers": workers}


_CPU_INPUT = {}


def _cpu_he(budget_s, worker):
    """Harmonic Energy, vectorised over 128-frame chunks of the bench signal (numpy.fft.rfft, float64)."""
    from oracle import harmonic_energy as o_he
    x = _CPU_INPUT["he"]
    nfr = (x.shape[0] - N_FFT) // HOP + 1
    chunk, done, t0, f = 128, 0, time.perf_counter(), worker * 7
    while time.perf_counter() - t0 < budget_s:
        lo = (f % (nfr // chunk)) * chunk * HOP
        o_he.he_frames(x[lo:lo + (chunk - 1) * HOP + N_FFT], FS, N_FFT, HOP)
        done += chunk
        f += 1
    return done, time.perf_counter() - t0


def _cpu_esacf(budget_s, worker, frame, hop, fs):
    import warnings
    from oracle import esacf as o_esacf
    x = _CPU_INPUT["he"] if fs == FS and hop != frame else _CPU_INPUT["clip44"]
    per = 4
    done, t0, f = 0, time.perf_counter(), worker * 3
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        while time.perf_counter() - t0 < budget_s:
            lo = (f * per * hop) % max(1, x.shape[0] - ((per - 1) * hop + frame))
            o_esacf.esacf_frames(x[lo:lo + (per - 1) * hop + frame], fs, frame_size=frame, hop=hop)
            done += per
            f += 1
    return done, time.perf_counter() - t0


def _cpu_esacf_clips(budget_s, worker):
    fs = CFG["esacf_fs"]
    return _cpu_esacf(budget_s, worker, int(fs * 46.4 / 1000), int(fs * 46.4 / 1000), fs)


def _cpu_esacf_stft(budget_s, worker):
    return _cpu_esacf(budget_s, worker, N_FFT, HOP, FS)


def _cpu_corpus(budget_s, worker):
    """All four methods on whole 2 s clips @22.05 kHz, one clip after the other like the reference's tests/test.py loop."""
    import warnings
    from oracle import esacf as o_esacf, harmonic_energy as o_he, iterative_f0 as o_if0, prime_multif0 as o_prime
    clips, fs = _CPU_INPUT["clips22"], CFG["corpus_fs"]
    done, t0 = 0, time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        while time.perf_counter() - t0 < budget_s:
            x = clips[(worker + done) % clips.shape[0]]
            o_esacf.esacf_compute(x, fs)
            o_he.he_compute(x, fs)
            o_if0.iterative_f0_compute(x, fs)
            o_prime.prime_compute(x, fs)
            done += 1
    return done, time.perf_counter() - t0


def _cpu_if0(budget_s, worker):
    """Iterative-F0 on 4-frame pieces (32768 samples) of a 44.1 kHz stream; unit = seconds of audio."""
    import warnings
    from oracle import iterative_f0 as o_if0
    x, fs, nf = _CPU_INPUT["he"], CFG["stream_fs"], CFG["if0_frame"]
    piece = 4 * nf
    done, t0, f = 0, time.perf_counter(), worker
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        while time.perf_counter() - t0 < budget_s:
            lo = (f * piece) % max(1, x.shape[0] - piece)
            o_if0.iterative_f0_compute(x[lo:lo + piece], fs, frame_size=nf)
            done += piece / fs
            f += 1
    return done, time.perf_counter() - t0


_CPU_LEGS = {"he": _cpu_he, "esacf_clips": _cpu_esacf_clips, "esacf_stft": _cpu_esacf_stft, "corpus": _cpu_corpus,
             "if0": _cpu_if0}


def _cpu_worker(arg):
    name, budget, worker = arg
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)          # one thread per worker
    except Exception:
        pass
    return _CPU_LEGS[name](budget, worker)


def cpu_baselines(budget_s=6.0, legs=("he", "esacf_clips", "esacf_stft", "corpus", "if0")):
    """Every leg twice: one process on one core, then one process per physical core (multiprocessing, fork): units
    done / wall clock of the slowest worker.  Bounded samples of the same workloads the GPU legs run."""
    import multiprocessing as mp
    info = host_cpu_info()
    _CPU_INPUT["he"] = synth_signal(20260101, frames=2048)
    _CPU_INPUT["clips22"] = synth_clips_numpy(16, CFG["corpus_fs"], 2.0)
    _CPU_INPUT["clip44"] = synth_clips_numpy(4, CFG["esacf_fs"], 2.0).reshape(-1)
    from oracle import esacf, harmonic_energy, iterative_f0, prime_multif0  # noqa: F401  (imported before the fork)
    units = {"he": "frames/s", "esacf_clips": "frames/s", "esacf_stft": "frames/s", "corpus": "clips/s",
             "if0": "x real time"}
    samples = {
        "he": "N=4096 hop=1024 frames of a 47 s stretch of the bench signal through oracle/harmonic_energy.py (numpy.fft.rfft, float64, 128 frames per call)",
        "esacf_clips": "46.4 ms frames of 44.1 kHz polyphonic clips through oracle/esacf.py (4 frames per call)",
        "esacf_stft": "N=4096 hop=1024 frames of the bench signal through oracle/esacf.py (phase-vocoder regime, 4 frames per call)",
        "corpus": "2 s clips @22.05 kHz through all four oracle methods, one clip after the other",
        "if0": "32768-sample pieces of a 44.1 kHz signal through oracle/iterative_f0.py",
    }
    out = {}
    ctx = mp.get_context("fork")
    for name in legs:
        done1, el1 = _cpu_worker((name, budget_s, 0))
        rec = {"value": done1 / el1, "unit": units[name], "cores": 1, "kind": "port",
               "sample": "%s; %.1f s on 1 core" % (samples[name], el1)}
        p = info["workers"]   # one per physical core the container may actually use
        if p > 1:
            t0 = time.perf_counter()
            with ctx.Pool(p) as pool:
                res = pool.map(_cpu_worker, [(name, budget_s, w) for w in range(p)], chunksize=1)
            wall = time.perf_counter() - t0   # includes the fork and the slowest worker
            rec["all_cores"] = {"value": sum(r[0] for r in res) / max(max(r[1] for r in res), 1e-9), "cores": p,
                                "wall_s": wall}
        out[name] = rec
    # the reference's own loop structure for the headline path (one frame per call), for honesty
    from oracle import harmonic_energy as o_he
    x = _CPU_INPUT["he"]
    t1, k = time.perf_counter(), 256
    for i in range(k):
        o_he.he_frames(x[i * HOP:i * HOP + N_FFT], FS, N_FFT)
    out["he"]["per_frame_loop_frames_per_s"] = k / (time.perf_counter() - t1)
    for rec in out.values():
        rec["host"] = info
    import numpy
    out["he"]["numpy"] = numpy.__version__
    return out


# ---------------------------------------------------------------------------------------------------------------
# algorithmic work per unit of each kernel (DESIGN.md section 5 states the same figures)
# ---------------------------------------------------------------------------------------------------------------
def kernel_models(n, mh, channels=70, nf=8192):
    """name -> (bytes per unit, flops per unit, unit): compulsory HBM bytes and textbook flop counts."""
    lg = math.log2(max(n, 2))
    return {
        # ESACF, unit = frame of n samples, mh = (n-1)//2 lags
        "bandsplit_kernel": (4 * n + 16 * n, 90 * n, "frame"),            # fp32 in, (x_lo, x_hi) fp64 out; 12 all-pass + 13-tap FIR + 3 biquads
        "sacf_kernel": (16 * n + 8 * mh, 2 * 5 * n * lg + 40 * n, "frame"),   # two n-point complex DFTs + |.|^0.67 (log+exp) per bin
        "sacf_big_kernel": (16 * n + 8 * mh, 2 * 5 * n * lg + 40 * n, "frame"),
        "sacf_pfa_kernel": (16 * n + 8 * mh, 2 * 5 * n * lg + 40 * n, "frame"),   # the same algorithmic count whatever the engine
        "pv_enhance_kernel": (16 * mh, 2 * 6 * 2.5 * 2048 * 11, "frame"),  # two real vocoder rates x (4 STFT + 2 ISTFT) 2048-point real FFTs
        "peakpick_kernel": (8 * mh, 4 * mh, "frame"),
        "scatter_kernel": (96, 0, "frame"),
        # Iterative-F0, unit = sample (front end) or frame (spectra, search)
        # COMPULSORY bytes: the samples in once (4 B) and 12 doubles out per frame; what the three kernels hand
        # each other through HBM -- 8 B x channels per sample from the front end to the spectra, the 2 nf-bin summary
        # spectrum to the period search -- is INTERMEDIATE traffic and listed separately (INTERMEDIATE_BYTES below)
        "if0_frontend_kernel": (4, 110 * channels, "sample"),   # 17 IIR stages + 13-tap FIR per channel and sample
        "if0_spectrum_kernel": (0, channels * (2.5 * 2 * nf * math.log2(2 * nf) + nf), "frame"),
        "if0_periodicity_kernel": (96, 0, "frame"),
    }


# profile marks of the library (mpx_profile_*) -> kernels they cover, as the profiler names them
MARK_KERNELS = {"prime_kernel": ("prime_pers_kernel", "prime_kernel"),
                "if0_frontend_kernel": ("if0_frontend_kernel", "if0_frontend2_kernel"),   # pipelined | sequential (mpx_if0.hip)
                "if0_spectrum_kernel": ("if0_spectrum_split_kernel", "if0_spectrum_dif_kernel", "if0_spectrum_kernel"),
                "he_kernel": ("he_wave_kernel", "he_kernel", "he_blue_kernel"),
                "coopfit_kernel": ("coopfit_kernel", "coopfit8_kernel")}   # four or eight fits to a wave, chosen on the device (mpx_esacf.hip)
_TRAFFIC = None


def measured_traffic(pmc_workload, mark):
    """HBM bytes of the launches one profile mark covers, inside one workload of scripts/pmc_workloads.py, from the last
    PMC collection (profiles/traffic_latest.json; bench.py cannot run rocprofv3 on itself).  (bytes, note) or (None, None)."""
    global _TRAFFIC
    if _TRAFFIC is None:
        try:
            with open(os.path.join(ROOT, "profiles", "traffic_latest.json")) as fh:
                _TRAFFIC = json.load(fh)
        except Exception:
            _TRAFFIC = {}
    ks = _TRAFFIC.get("kernels", {}).get(pmc_workload, {})
    total, hit = 0.0, []
    for name, shapes in ks.items():
        base = name.split("<")[0]
        if any(base == p for p in MARK_KERNELS.get(mark, (mark,))):
            # a mark covers every launch shape of its kernels in the call (Prime-multiF0: one launch per chirp-z class)
            total += sum(sh["bytes_per_launch"] for sh in shapes)
            hit.append(name)
    if not hit:
        return None, None
    return total, "round %s PMC, %s: %s" % (_TRAFFIC.get("round"), pmc_workload, ", ".join(sorted(hit)))


def with_traffic(r, pmc_workload, mark, launches=1):
    """Fill roofline.traffic (bytes per call of the marked kernels x launches) and the ratio to the compulsory bytes."""
    if r is None:
        return r
    marks = mark.split("+")
    tot, notes = 0.0, []
    for m in marks:
        b, note = measured_traffic(pmc_workload, m)
        if b is None:
            return r
        tot += b
        notes.append(note)
    r["traffic"] = tot * launches
    r["traffic_note"] = "HBM-side bytes (calibrated factor x FETCH_SIZE + WRITE_SIZE, scripts/pmc_to_traffic.py) of these launches; " + "; ".join(notes)
    comp = r.get("bytes_per_unit", 0) * r.get("units_per_launch", 0)
    if r.get("compulsory_bytes"):
        comp = r["compulsory_bytes"]
    if comp:
        r["compulsory_bytes"] = comp
        r["wasted_traffic_ratio"] = r["traffic"] / comp
    ib = r.get("intermediate_bytes_per_unit", 0) * r.get("units_per_launch", 0)
    if ib:   # a kernel that hands data to the next one of its method: the measured bytes against compulsory + hand-off
        r["traffic_vs_compulsory_plus_intermediate"] = r["traffic"] / (comp + ib)
    return r


def he_kernel_name(f32):
    """The dominant kernel of the headline step: fp64 4096-sample frames run the wave-per-frame kernel (csrc/mpx_he_wave.hpp),
    fp32 the workgroup-per-frame one."""
    return "he_kernel<4096,256,float>" if f32 else "he_wave_kernel<8,4>"


def intermediate_bytes(name, channels=70, nf=8192):
    """Bytes per unit a kernel moves through HBM that are NOT compulsory: hand-offs between the kernels of one method."""
    return {"if0_frontend_kernel": 8 * channels,                      # writes [channel][t] fp64 for the spectra
            "if0_spectrum_kernel": 8 * nf * channels + 16 * nf,        # reads it back, writes the 2 nf-bin summary spectrum
            "if0_periodicity_kernel": 3 * 16 * nf}.get(name, 0)        # summary spectrum in, residual / detected spectra


def roofline_of(name, ms, units, model):
    """Both roofs for one kernel; `bound` is the one it sits closer to."""
    b, f, unit = model
    hbm = b * units / (ms * 1e-3) if ms > 0 else 0.0
    fl = f * units / (ms * 1e-3) if ms > 0 else 0.0
    hbm_frac, valu_frac = hbm / HBM_PEAK, fl / F64_PEAK
    if f and valu_frac >= hbm_frac:
        r = {"bound": "valu_f64", "achieved": fl / 1e12, "peak": F64_PEAK / 1e12, "unit": "TFLOP/s", "frac": valu_frac}
    else:
        r = {"bound": "hbm", "achieved": hbm / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": hbm_frac}
    r.update({"kernel": name, "kernel_ms": ms, "units_per_launch": units, "unit_of_work": unit,
              "bytes_per_unit": b, "flops_per_unit": f, "hbm_frac": hbm_frac, "valu_f64_frac": valu_frac,
              "traffic": None})
    ib = intermediate_bytes(name)
    if ib:
        r["intermediate_bytes_per_unit"] = ib
        r["hbm_frac_with_intermediate"] = (b + ib) * units / (ms * 1e-3) / HBM_PEAK if ms > 0 else 0.0
    return r


FLOPS_PER_FIT_EVAL = 21 * 35   # one MINPACK function evaluation of a gaussian peak fit: 21 residuals x (exp ~30 flops + 5)


def fit_roofline(kms, stats):
    """The two gaussian-fit kernels together (peakfit_kernel runs the fits, coopfit_kernel finishes the runaway ones): work
    counted in MINPACK function evaluations (mpx_esacf_fit_stats) x 735 flops for the model evaluation alone -- the QR of
    the 21 x 3 jacobian and the 3 x 3 trust-region algebra of an iteration come on top and are not counted -- against the
    fp64 vector peak.  Bytes: the 21-sample window (168 B) in, 12 B out per fit: nothing."""
    ms = kms.get("peakfit_kernel", 0.0) + kms.get("coopfit_kernel", 0.0)
    fl = stats["evaluations"] * FLOPS_PER_FIT_EVAL / (ms * 1e-3) if ms > 0 else 0.0
    hbm = stats["fits"] * 180 / (ms * 1e-3) if ms > 0 else 0.0
    return {"bound": "valu_f64", "achieved": fl / 1e12, "peak": F64_PEAK / 1e12, "unit": "TFLOP/s", "frac": fl / F64_PEAK,
            "kernel": "peakfit_kernel+coopfit_kernel", "kernel_ms": ms, "units_per_launch": stats["evaluations"],
            "unit_of_work": "function evaluation", "bytes_per_unit": 0, "flops_per_unit": FLOPS_PER_FIT_EVAL,
            "compulsory_bytes": stats["fits"] * 180,   # a fit's 21-sample window in (168 B), centre + flag out (12 B)
            "hbm_frac": hbm / HBM_PEAK, "valu_f64_frac": fl / F64_PEAK, "traffic": None, "fits": stats["fits"],
            "evaluations_per_fit": stats["evaluations"] / max(stats["fits"], 1), "fits_finished_cooperatively": stats["parked"]}


def dominant(prof):
    name = max(prof, key=lambda k: prof[k][1])
    return name, prof[name][1] / max(prof[name][0], 1)


# ---------------------------------------------------------------------------------------------------------------
COMPACT_LIMIT = 6000   # bytes of the last stdout line (tests/test_gpu_bench_contract.py asserts it)


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _short(s, n=120):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 1] + "~"


def _r(v, sig=6):
    """numbers to `sig` significant digits (the full record keeps every bit)"""
    if isinstance(v, float):
        return float("%.*g" % (sig, v)) if math.isfinite(v) else None
    if isinstance(v, dict):
        return {k: _r(x, sig) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, sig) for x in v]
    return v


ROOF_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "compulsory_bytes",
             "wasted_traffic_ratio")


def _cpu_compact(cb):
    if not cb:
        return None
    r = _pick(cb, ("value", "unit", "cores", "kind"))
    r["sample"] = _short(cb.get("sample", ""), 100)
    if isinstance(cb.get("all_cores"), dict):
        r["all_cores"] = _pick(cb["all_cores"], ("value", "cores"))
    return r


def compact_record(out, full_path=None):
    """The driver's line: the contract's keys, the headline roofline + cpu_baseline, and one short entry per workload."""
    rec = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                      "vs_baseline", "dtype", "data", "engine", "ms_per_step_repeats", "value_one_in_flight",
                      "ms_per_step_one_in_flight"))
    cfg = out.get("config", {})
    rec["config"] = _pick(cfg, ("frames_per_gpu", "fft", "hop", "fs", "repeats", "repeat_statistic", "batches_in_flight",
                                "distinct_input_signals"))
    rec["config"]["workload"] = _short(cfg.get("workload", ""), 160)
    roof = out.get("roofline", {})
    rec["roofline"] = _pick(roof, ROOF_KEYS + ("bytes_per_frame", "frames_per_launch", "step_ms_hip_events"))
    if "secondary" in roof:
        rec["roofline"]["secondary"] = _pick(roof["secondary"], ("bound", "achieved", "peak", "unit", "frac"))
    if out.get("cpu_baseline"):
        rec["cpu_baseline"] = _cpu_compact(out["cpu_baseline"])
    wl = {}
    for name, w in (out.get("workloads") or {}).items():
        e = _pick(w, ("value", "unit", "scaling", "value_definition", "value_warm", "value_first_pass", "value_cold",
                      "value_three_in_flight", "value_with_streaming_synthesis", "value_without_synthesis",
                      "oracle_spot_check"))
        ms = w.get("ms_per_batch", 1e3 * w["wall_s"] if "wall_s" in w else None)
        if ms is not None:
            e["ms"] = ms
        r = w.get("roofline") or {}
        e.update(_pick(r, ("kernel", "kernel_ms", "bound", "frac", "traffic", "compulsory_bytes", "wasted_traffic_ratio")))
        if "hbm_frac_whole_path" in w:
            e["hbm_frac_whole_path"] = w["hbm_frac_whole_path"]
        km = w.get("kernels_ms") or w.get("kernels_ms_total")
        if km:
            e["kernels_ms"] = {k: v for k, v in sorted(km.items(), key=lambda kv: -kv[1])[:6]}
        cb = w.get("cpu_baseline")
        if cb:
            e["cpu"] = _pick(cb, ("value", "cores"))
            if isinstance(cb.get("all_cores"), dict):
                e["cpu"]["all_cores"] = _pick(cb["all_cores"], ("value", "cores"))
        e["workload"] = _short((w.get("config") or {}).get("workload", ""), 110)
        wl[name] = e
    if wl:
        rec["workloads"] = wl
    if full_path:
        rec["full_record"] = full_path
    return _r(rec)


def write_full_record(out, path):
    """the uncut record: `path`, and a copy under gpurun_out/ when that directory exists (it travels back from the GPU box)"""
    written = None
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (path, os.path.join(here, "gpurun_out", os.path.basename(path))):
        try:
            if p != path and not os.path.isdir(os.path.dirname(p)):
                continue
            with open(p, "w") as f:
                json.dump(out, f)
            written = written or os.path.relpath(p, os.getcwd())
        except OSError:
            pass
    return written


def main():
    # tests only: a stand-in for torch.cuda + the HIP engine, so that the world > 1 code below runs over gloo on CPU
    stub = importlib.import_module(os.environ["MPX_BENCH_STUB"]) if os.environ.get("MPX_BENCH_STUB") else None
    if stub is not None:
        stub.configure(sys.modules[__name__])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 0.1 s of GPU work.  The device needs on the order of 0.05 s under load to reach its sustained clock; the
    # untimed pre-heat below makes the figure independent of K and W.
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--streams", type=int, default=4,
                    help="batches in flight: step i goes to context/stream i %% S (1 = strictly one launch after the other)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the K-step timed loop is run this many times, each bracketed by barrier + synchronize; the MEDIAN is "
                         "reported (the driver's short run, K = 20, is 0.8 ms of GPU time: one sample of it is noise)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="skip the secondary workloads")
    ap.add_argument("--workloads", default="esacf_clips_4096,esacf_stft_8192,corpus_4096_all_methods,if0_stream_1h")
    ap.add_argument("--signals", type=int, default=NSIG, help="distinct input signals the steps rotate over")
    ap.add_argument("--f32", action="store_true", help="opt-in fp32 engine (not the headline)")
    ap.add_argument("--full-json", default="bench_full.json",
                    help="file that receives the full record; stdout's last line is the compact one (<= 6 KB)")
    args = ap.parse_args()

    # Nothing in the environment may change what is measured: the release library reads no development switches, and this
    # script refuses to run with any MPX_* variable other than its own two set (MPX_LIB_PATH would swap the library,
    # MPX_DETERMINISTIC the fit scheduling).
    foreign = sorted(k for k in os.environ if k.startswith("MPX_") and k not in ("MPX_BENCH_STUB", "MPX_BENCH_CPU_BUDGET"))
    if foreign:
        sys.exit("bench.py: unset %s (bench numbers are taken with the default library and defaults only)" % ", ".join(foreign))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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

    if stub is None:
        from chord_detection_amd import _lib
        if _lib.load().mpx_dev_knobs():
            sys.exit("bench.py: the loaded library was built with -DMPX_DEV_KNOBS (development switches); use the release build")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        make_engine = lambda: cd.Engine(local_rank, f32=args.f32)
        backend = "nccl"
    else:
        dev, make_engine, backend = torch.device("cpu"), stub.Engine, "gloo"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if stub is None:
            dist.init_process_group(backend=backend, device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    def dev_sync():
        if dev.type == "cuda":
            torch.cuda.synchronize()

    def barrier():
        if world > 1:
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
    sigs = [synth_signal_device(20260101 + 1000 * rank + k, dev, FRAMES) for k in range(nsig)]
    x_host = sigs[0].cpu().numpy()
    n = sigs[0].numel()
    steps, warmup = args.steps, args.warmup
    d_frames = torch.empty((FRAMES, 12), dtype=torch.float64, device=dev)
    d_sums = torch.zeros((max(steps, warmup, nstreams, nsig, 1), 12), dtype=torch.float64, device=dev)
    dev_sync()

    def step(i, e=None):
        # the product path for one signal: chroma summed over frames, no per-frame rows
        (e or engs[i % nstreams]).harmonic_energy_dev(sigs[i % nsig].data_ptr(), n, FS, N_FFT, HOP, None,
                                                     d_sums.data_ptr() + i * 96)

    def sync_engines():
        for e in engs:
            e.synchronize()

    # Untimed pre-heat: the same launches for PREHEAT_MS of wall time, so that the device is at its sustained clock
    # whatever W and K are (measured: 62.3 us/step with W=20, K=200 from a cold device, 55.4 at any larger K).
    t_pre = time.perf_counter()
    while 1e3 * (time.perf_counter() - t_pre) < PREHEAT_MS:
        for j in range(64):
            step(j % max(nstreams, nsig))
        sync_engines()
    for i in range(warmup):
        step(i)
    if world > 1:  # the job's one collective, once untimed: RCCL sets its rings up on first use
        sync_engines()
        dist.all_gather([torch.empty_like(d_sums) for _ in range(world)], d_sums)
    # The timed region, `repeats` times: EXACTLY K steps between barrier + synchronize on both sides, the rank maximum of
    # each repeat, and the median of the repeats is what `value` is computed from (every repeat is listed in the output).
    repeats = max(1, args.repeats)
    rep_s, host_enqueue_ms, gathered = [], 0.0, None
    for rep in range(repeats):
        barrier()
        t0 = time.perf_counter()
        th0 = time.perf_counter()
        for i in range(steps):
            step(i)
        host_enqueue_ms = 1e3 * (time.perf_counter() - th0) / max(steps, 1)
        if world > 1:
            sync_engines()
            gathered = [torch.empty_like(d_sums) for _ in range(world)]
            dist.all_gather(gathered, d_sums)          # one RCCL gather of the 12-vectors, at the end
        else:
            sync_engines()
        barrier()
        el = time.perf_counter() - t0
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rep_s.append(float(t.item()))
    elapsed = sorted(rep_s)[len(rep_s) // 2]
    sums = d_sums[:steps].cpu().numpy()

    # the same K steps strictly one launch after the other on ONE context/stream (what the per-launch roofline describes)
    one_s = []
    for rep in range(repeats):
        barrier()
        t1 = time.perf_counter()
        for i in range(steps):
            step(i, eng)
        eng.synchronize()
        barrier()
        one = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(one, op=dist.ReduceOp.MAX)
        one_s.append(float(one.item()))
    elapsed_one = sorted(one_s)[len(one_s) // 2]

    # HIP events on that stream: the full step (with the in-kernel reduction) ...
    reps = max(min(steps, 2000), 50)
    d_seq = torch.zeros(12, dtype=torch.float64, device=dev)
    eng.timer_begin()
    for r in range(reps):
        eng.harmonic_energy_dev(sigs[r % nsig].data_ptr(), n, FS, N_FFT, HOP, None, d_seq.data_ptr())
    step_ms_events = eng.timer_end() / reps
    # ... and the dominant kernel alone (per-frame rows out, no final 12-vector reduction): the roofline's launch duration
    eng.synchronize()
    reps = max(steps, 50)
    eng.timer_begin()
    for r in range(reps):
        eng.harmonic_energy_dev(sigs[r % nsig].data_ptr(), n, FS, N_FFT, HOP, d_frames.data_ptr(), None)
    kern_ms = eng.timer_end() / reps
    achieved = B_ALG * FRAMES / (kern_ms * 1e-3)

    # sanity: the benchmarked output is the real thing (checked against the oracle on a few frames)
    from oracle import harmonic_energy as o_he
    eng.harmonic_energy_dev(sigs[0].data_ptr(), n, FS, N_FFT, HOP, d_frames.data_ptr(), None)
    eng.synchronize()
    got = d_frames[:4].cpu().numpy()
    want = o_he.he_frames(x_host[:3 * HOP + N_FFT], FS, N_FFT, HOP)
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
                    dev_sync=dev_sync, stub=stub, cpu=cpu)
        for name in want_w:
            barrier()
            rec = WORKLOADS[name](ctxw)
            if rec is not None:
                workloads[name] = rec

    if rank == 0:
        total_frames = FRAMES * world * steps
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
            "value_one_in_flight": total_frames / elapsed_one,
            "ms_per_step_one_in_flight": 1e3 * elapsed_one / max(steps, 1),
            "config": {"workload": "Harmonic Energy STFT->chromagram, 8192 synthetic 44.1 kHz frames per GPU, "
                                   "4096-pt FFT hop 1024 (BASELINE.json configs[1])",
                       "frames_per_gpu": FRAMES, "fft": N_FFT, "hop": HOP, "fs": FS, "untimed_preheat_ms": PREHEAT_MS,
                       "repeats": repeats, "repeat_statistic": "median",
                       "batches_in_flight": nstreams, "distinct_input_signals": nsig,
                       "input_bytes_rotated_over": int(nsig * n * 4),
                       "sharding": "frames per rank, no data-path collective; one RCCL all_gather of 12-vectors at the end"},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": traffic,
                         "traffic_note": "bytes/launch; %s; algorithmic = %d" % (traffic_note, B_ALG * FRAMES),
                         "compulsory_bytes": B_ALG * FRAMES,
                         "wasted_traffic_ratio": (traffic / (B_ALG * FRAMES)) if traffic else None,
                         "kernel": kname,
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
    if world > 1:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------
# secondary workloads: each returns a record on rank 0 (None elsewhere)
# ---------------------------------------------------------------------------------------------------------------
def _max_over_ranks(c, seconds):
    t = c["torch"].tensor([seconds], dtype=c["torch"].float64, device=c["dev"])
    if c["world"] > 1:
        c["dist"].all_reduce(t, op=c["dist"].ReduceOp.MAX)
    return float(t.item())


def _cpu_rec(c, leg):
    return c["cpu"][leg] if c["cpu"] is not None else None


def wl_esacf_clips(c):
    """configs[2]: every rank runs its own 4096 clips (weak scaling) through the batch entry point of the C ABI, the clips
    resident in HBM; per-clip framing (44 frames of 2046 samples per 2 s clip @44.1 kHz)."""
    torch, np, eng = c["torch"], c["np"], c["eng"]
    from chord_detection_amd import corpus
    fs, secs, clips = CFG["esacf_fs"], CFG["esacf_clip_seconds"], CFG["esacf_clips"]
    frame = int(fs * 46.4 / 1000)
    uniq = corpus.synth_chunk(list(range(64 * c["rank"], 64 * c["rank"] + min(64, clips))), fs, secs, c["dev"])
    x = uniq.repeat((clips + uniq.shape[0] - 1) // uniq.shape[0], 1)[:clips].contiguous()
    per_clip = -(-x.shape[1] // frame)
    frames = clips * per_clip
    first = eng.esacf_batch(x, fs, frame)          # plans, workspaces
    reps = 3
    c["barrier"]()
    t0 = time.perf_counter()
    for _ in range(reps):
        got = eng.esacf_batch(x, fs, frame)
    c["barrier"]()
    wall = _max_over_ranks(c, (time.perf_counter() - t0) / reps)
    eng.profile_begin()
    eng.esacf_batch(x, fs, frame)
    prof = eng.profile_end()
    stats = eng.esacf_fit_stats() if hasattr(eng, "esacf_fit_stats") else None
    if not np.array_equal(got, first):
        sys.exit("bench: ESACF batch results differ between runs (non-deterministic)")
    ok = None
    if c["stub"] is None:   # spot check against the oracle
        import warnings
        from oracle import esacf as o_esacf
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = o_esacf.esacf_compute(x[0, :3 * frame].cpu().numpy(), fs)
        ok = bool(np.allclose(eng.esacf(x[0, :3 * frame].cpu().numpy(), fs, frame), want, rtol=1e-5, atol=1e-12))
    if c["rank"] != 0:
        return None
    kms = {k: v[1] / max(v[0], 1) for k, v in prof.items()}
    models = kernel_models(frame, (frame - 1) // 2)
    dom, dms = dominant(prof)
    rec = {"value": frames * c["world"] / wall, "unit": "frames/s", "clips_per_s": clips * c["world"] / wall,
           "ms_per_batch": 1e3 * wall, "scaling": "weak", "dtype": "f64",
           "config": {"workload": "ESACF, %d clips x %.0f s @%d Hz per GPU, %d-sample frames (BASELINE.json configs[2])"
                                  % (clips, secs, fs, frame), "frames_per_gpu": frames, "entry": "mpx_esacf_batch, clips in HBM"},
           "kernels_ms": kms, "kernel_ms_total": sum(kms.values()), "oracle_spot_check": ok,
           "roofline": roofline_of(dom, dms, frames, models[dom]) if dom in models else
           (fit_roofline(kms, stats) if stats else {"kernel": dom, "kernel_ms": dms}),
           "rooflines": {k: roofline_of(k, ms, frames, models[k]) for k, ms in kms.items() if k in models}}
    if stats:
        rec["rooflines"]["peakfit_kernel+coopfit_kernel"] = fit_roofline(kms, stats)
    for k, r in rec["rooflines"].items():
        with_traffic(r, "esacf_clips", k)
    with_traffic(rec["roofline"], "esacf_clips", rec["roofline"]["kernel"])
    rec["hbm_frac_whole_path"] = (4 * frame + 48) * rec["value"] / c["world"] / HBM_PEAK
    if _cpu_rec(c, "esacf_clips"):
        rec["cpu_baseline"] = _cpu_rec(c, "esacf_clips")
    return rec


def wl_esacf_stft(c):
    """The north star's Target: STFT -> ESACF -> chromagram on ONE signal of 8192 overlapping frames (N=4096, hop 1024),
    device-resident (mpx_esacf_dev); one launch sequence at a time, and with three batches in flight on three contexts."""
    torch, np, eng = c["torch"], c["np"], c["eng"]
    sigs, n = c["sigs"], c["sigs"][0].numel()
    nf = FRAMES
    outs = [(torch.zeros((nf, 12), dtype=torch.float64, device=c["dev"]), torch.zeros(12, dtype=torch.float64, device=c["dev"]))
            for _ in range(3)]
    engs = (c["engs"] + [c["make_engine"]() for _ in range(3)])[:3]
    for e, (fr, sm) in zip(engs, outs):
        e.esacf_dev(sigs[0].data_ptr(), n, FS, N_FFT, HOP, fr.data_ptr(), sm.data_ptr())
        e.synchronize()
    reps = 6
    c["barrier"]()
    t0 = time.perf_counter()
    for r in range(reps):
        eng.esacf_dev(sigs[r % len(sigs)].data_ptr(), n, FS, N_FFT, HOP, outs[0][0].data_ptr(), outs[0][1].data_ptr())
    eng.synchronize()
    c["barrier"]()
    wall1 = _max_over_ranks(c, (time.perf_counter() - t0) / reps)
    c["barrier"]()
    t0 = time.perf_counter()
    for r in range(3 * reps):
        e, (fr, sm) = engs[r % 3], outs[r % 3]
        e.esacf_dev(sigs[r % len(sigs)].data_ptr(), n, FS, N_FFT, HOP, fr.data_ptr(), sm.data_ptr())
    for e in engs:
        e.synchronize()
    c["barrier"]()
    wall3 = _max_over_ranks(c, (time.perf_counter() - t0) / (3 * reps))
    eng.profile_begin()
    eng.esacf_dev(sigs[0].data_ptr(), n, FS, N_FFT, HOP, outs[0][0].data_ptr(), outs[0][1].data_ptr())
    eng.synchronize()
    prof = eng.profile_end()
    stats = eng.esacf_fit_stats() if hasattr(eng, "esacf_fit_stats") else None
    ok = None
    if c["stub"] is None:
        import warnings
        from oracle import esacf as o_esacf
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = o_esacf.esacf_frames(c["x_host"][:2 * HOP + N_FFT].astype(np.float64), FS, N_FFT, HOP)
        ok = bool(np.allclose(outs[0][0][:3].cpu().numpy(), want[:3], rtol=1e-5, atol=1e-12))
    if c["rank"] != 0:
        return None
    kms = {k: v[1] / max(v[0], 1) for k, v in prof.items()}
    models = kernel_models(N_FFT, (N_FFT - 1) // 2)
    dom, dms = dominant(prof)
    rec = {"value": nf * c["world"] / wall1, "unit": "frames/s", "ms_per_batch": 1e3 * wall1,
           "value_three_in_flight": nf * c["world"] / wall3, "ms_per_batch_three_in_flight": 1e3 * wall3,
           "scaling": "weak", "dtype": "f64",
           "config": {"workload": "STFT->ESACF->chromagram, one signal of 8192 frames per GPU, N=4096 hop 1024 @44.1 kHz "
                                  "(BASELINE.json north_star Target)", "frames_per_gpu": nf, "entry": "mpx_esacf_dev"},
           "kernels_ms": kms, "kernel_ms_total": sum(kms.values()), "oracle_spot_check": ok,
           "roofline": roofline_of(dom, dms, nf, models[dom]) if dom in models else
           (fit_roofline(kms, stats) if stats else {"kernel": dom, "kernel_ms": dms, "bound": "latency", "frac": None, "traffic": None}),
           "rooflines": {k: roofline_of(k, ms, nf, models[k]) for k, ms in kms.items() if k in models},
           "hbm_frac_whole_path": B_ALG * nf / wall1 / HBM_PEAK,
           "hbm_frac_whole_path_three_in_flight": B_ALG * nf / wall3 / HBM_PEAK}
    if stats:
        rec["rooflines"]["peakfit_kernel+coopfit_kernel"] = fit_roofline(kms, stats)
    for k, r in rec["rooflines"].items():
        with_traffic(r, "esacf_stft", k)
    with_traffic(rec["roofline"], "esacf_stft", rec["roofline"]["kernel"])
    if _cpu_rec(c, "esacf_stft"):
        rec["cpu_baseline"] = _cpu_rec(c, "esacf_stft")
    return rec


def wl_corpus(c):
    """configs[3]: all four methods over 4096 clips per GPU, clip-sharded, ONE gather of the 12-vectors (the corpus driver)."""
    np = c["np"]
    from chord_detection_amd import corpus
    per, world, rank = CFG["corpus_clips_per_gpu"], c["world"], c["rank"]
    fs = CFG["corpus_fs"]
    kw = {}
    if c["stub"] is not None:
        kw = {"compute": c["stub"].corpus_compute}
    dev = c["dev"] if c["dev"].type == "cuda" else None
    # untimed pass over one full-size chunk: the contexts' grow-only workspaces reach their final size here (Iterative-F0's
    # front end alone is 25 GB for 1024 clips; the first allocation of that size on a fresh box takes over a second)
    corpus.run_corpus(min(per, 1024) * world, (1, 2, 3, 4), fs, 2.0, 1024, rank, world, c["local_rank"], synth_device=dev, **kw)
    # The corpus is in HBM when the clock starts (0.7 GB per GPU): synthesising it inside the timed region, as a stand-in
    # for a decoder, cost 28 % of the GPU's time in round 2's figure (torch kernels next to the engines').  The driver's own
    # streaming mode (scripts/run_corpus.py) is timed as well and reported as `value_with_streaming_synthesis`.
    t_s = time.perf_counter()
    block_in = corpus.synth_block(per * world, fs, 2.0, 1024, rank, world, synth_device=dev)
    synth_s = time.perf_counter() - t_s
    c["barrier"]()
    t0 = time.perf_counter()
    corpus.run_corpus(per * world, (1, 2, 3, 4), fs, 2.0, 1024, rank, world, c["local_rank"], synth_device=dev, **kw)
    c["barrier"]()
    wall_streaming = _max_over_ranks(c, time.perf_counter() - t0)
    streaming_synth_wait = corpus.LAST_SYNTH_SECONDS
    profs = []
    if c["stub"] is None:
        profs = [c["cd"].get_engine(c["local_rank"]), corpus._second_engine(c["local_rank"]),
                 corpus._second_engine((c["local_rank"], 3))]   # the driver's three contexts: methods 1 + 2 | 3 | 4
        for e in profs:
            e.profile_begin()
    c["barrier"]()
    t0 = time.perf_counter()
    lo, hi, block, spent = corpus.run_corpus(per * world, (1, 2, 3, 4), fs, 2.0, 1024, rank, world, c["local_rank"],
                                             synth_device=dev, resident=block_in, **kw)
    chroma = corpus.gather_blocks(block, per * world, world, rank, c["dev"] if world > 1 and c["stub"] is None else None)
    c["barrier"]()
    wall = _max_over_ranks(c, time.perf_counter() - t0)
    prof = {}
    for e in profs:
        for k, v in e.profile_end().items():
            a = prof.get(k, (0, 0.0))
            prof[k] = (a[0] + v[0], a[1] + v[1])
    if rank != 0:
        return None
    assert chroma.shape == (per * world, 4, 12)
    ktot = {k: v[1] for k, v in prof.items()}
    rec = {"value": per * world / wall, "unit": "clips/s", "wall_s": wall, "scaling": "weak", "dtype": "f64",
           "config": {"workload": "all four methods over %d clips x 2 s @%d Hz per GPU, clip-sharded, one all_gather of "
                                  "[clips, 4, 12] (BASELINE.json configs[3]); the clips are resident in HBM when the clock starts"
                                  % (per, fs), "clips_per_gpu": per},
           "seconds_per_method_rank0": {str(m): s for m, s in zip((1, 2, 3, 4), spent)},
           # `value` since round 3: the clips are resident in HBM when the clock starts (the CPU leg times the same region:
           # pre-synthesised clips through the four methods).  Rounds 1-2 timed the driver's on-device synthesis too: that
           # figure is `value_with_streaming_synthesis`; the two are not like for like across rounds.
           "value_definition": "r3+: clips resident in HBM, synthesis untimed (r2's definition = value_with_streaming_synthesis)",
           "value_without_synthesis": per * world / wall, "synthesis_seconds_rank0": synth_s,
           "untimed_synthesis_seconds_rank0": synth_s,
           "value_with_streaming_synthesis": per * world / wall_streaming, "wall_s_with_streaming_synthesis": wall_streaming,
           "streaming_synthesis_wait_seconds_rank0": streaming_synth_wait,
           "kernels_ms_total": ktot, "nonzero_rows": int((np.abs(chroma).sum(axis=2) > 0).sum())}
    if ktot:
        dom = max(ktot, key=ktot.get)
        samples = per * int(round(2.0 * fs))
        frames1 = per * -(-int(round(2.0 * fs)) // int(fs * 46.4 / 1000))
        models = kernel_models(int(fs * 46.4 / 1000), (int(fs * 46.4 / 1000) - 1) // 2)
        units = {"if0_frontend_kernel": samples, "if0_spectrum_kernel": per * -(-int(round(2.0 * fs)) // 8192),
                 "if0_periodicity_kernel": per * -(-int(round(2.0 * fs)) // 8192)}
        # Prime-multiF0, unit = clip: 12 * num_octave candidate frequencies, each cuts the clip into frames of int(8/f*fs)
        # samples and takes a real FFT of every frame: 2.5 N log2 N flops and 4 N bytes per frame
        ncl = int(round(2.0 * fs))
        pf = pb = 0.0
        for note in range(12):
            for octave in (1, 2):
                nc = int(8.0 / (130.8127826502993 * 2.0 ** (note / 12.0) * octave) * fs)
                nfr = -(-ncl // nc)
                pf += nfr * 2.5 * nc * math.log2(nc)
                pb += nfr * 4 * nc
        # compulsory: the clip's samples in ONCE and 12 doubles out; every candidate frequency cuts the same
        # clip into its own frames, so the kernel READS it once per candidate (pb): listed as re-read bytes, mostly L2 / MALL hits
        models["prime_kernel"] = (4 * ncl + 96, pf, "clip")
        units["prime_kernel"] = per
        if dom in models:
            rec["roofline"] = roofline_of(dom, ktot[dom], units.get(dom, frames1), models[dom])
        else:
            rec["roofline"] = {"kernel": dom, "kernel_ms": ktot[dom], "bound": "latency", "frac": None, "traffic": None}
        rec["rooflines"] = {k: roofline_of(k, ms, units.get(k, frames1), models[k]) for k, ms in ktot.items() if k in models}
        # PMC counters were taken on ONE 1024-clip chunk of the driver (scripts/pmc_workloads.py): x chunks per GPU
        chunks = -(-per // 1024)
        pmc_wl = {"prime_kernel": "prime", "if0_frontend_kernel": "if0_clips", "if0_spectrum_kernel": "if0_clips",
                  "if0_periodicity_kernel": "if0_clips"}
        for k, r in list(rec["rooflines"].items()) + [(rec["roofline"].get("kernel"), rec["roofline"])]:
            # (the ESACF side was counted on 4096 clips at once)
            with_traffic(r, pmc_wl.get(k, "esacf_1023"), k, launches=chunks if k in pmc_wl else per / 4096.0)
        if "prime_kernel" in rec["rooflines"]:
            rec["rooflines"]["prime_kernel"]["reread_bytes_per_unit"] = pb
        if rec["roofline"].get("kernel") == "prime_kernel":
            rec["roofline"]["reread_bytes_per_unit"] = pb
        rec["kernels_ms_note"] = ("sums over the driver's three contexts, whose kernels overlap on the GPU: HIP-event times of "
                                  "kernels that share the machine (alone, per 1024-clip chunk: scripts/dev/prime_time.py, if0_time.py)")
    if _cpu_rec(c, "corpus"):
        rec["cpu_baseline"] = _cpu_rec(c, "corpus")
    return rec


def wl_if0_stream(c):
    """configs[4]: Iterative-F0 over ONE 1 h stream @44.1 kHz, its frames block-partitioned over the ranks with a
    run-in halo of stream.engine_warmup() samples (40960 for the default chain; strong scaling), one gather of [frames, 12] (the long-stream driver)."""
    torch = c["torch"]
    from chord_detection_amd import stream
    fs, secs, nf_size = CFG["stream_fs"], CFG["stream_seconds"], CFG["if0_frame"]
    world, rank, local = c["world"], c["rank"], c["local_rank"]
    n = int(round(secs * fs))
    total_frames = stream.num_frames(n, nf_size)
    warm = stream.WARMUP if c["stub"] is not None else stream.engine_warmup(fs, local, frame_size=nf_size)
    f0, f1, s0, s1, _ = stream.shard_window(n, nf_size, world, rank, warm)
    sdev = c["dev"] if c["dev"].type == "cuda" else None
    x = stream.synth_stream(s0, s1, fs, sdev)
    c["dev_sync"]()
    if c["dev"].type == "cuda":
        torch.cuda.empty_cache()   # the synthesis' cached blocks slow the engine's first large hipMalloc down
    kw = {}
    if c["stub"] is not None:
        kw = {"compute": c["stub"].stream_compute}

    def compute_block():
        if c["stub"] is None:
            return stream.run_stream_rank(lambda a, b: x[a - s0:b - s0], n, fs, rank, world, nf_size, local)[2]
        return stream.run_stream_shard(lambda a, b: x.numpy(), n, fs, rank, world, nf_size, local, **kw)[2]

    t0 = time.perf_counter()
    block = compute_block()           # first pass: grows the contexts' workspaces (tens of GB of hipMalloc)
    cold = time.perf_counter() - t0
    c["barrier"]()
    t0 = time.perf_counter()
    block = compute_block()
    frames = stream.gather_frames(block, total_frames, world, rank, c["dev"] if world > 1 and c["stub"] is None else None)
    c["barrier"]()
    wall = _max_over_ranks(c, time.perf_counter() - t0)
    cold = _max_over_ranks(c, cold)
    prof = {}
    if c["stub"] is None:   # kernel breakdown: one context, this rank's first <= 10 minutes
        e = c["cd"].get_engine(local)
        m = min(x.numel(), int(600 * fs))
        e.profile_begin()
        e.iterative_f0(x[:m], fs, frame_size=nf_size)
        prof = e.profile_end()
        prof_samples = m
    if rank != 0:
        return None
    assert frames.shape == (total_frames, 12)
    # A 1 h stream is a one-shot job: `value` is the FIRST pass of the process (the contexts' workspaces -- 13 GB since the
    # library runs the call in time slices, 90 GB in round 3 -- are allocated inside it); `value_warm` is the second pass.
    rec = {"value": secs / cold, "unit": "x real time", "wall_s": cold, "first_pass_wall_s": cold,
           "value_first_pass": secs / cold, "value_warm": secs / wall, "warm_wall_s": wall,
           "value_definition": "r4+: first pass of the process, workspace allocation included (r1-r3 reported the second pass: value_warm)",
           "scaling": "strong",
           "dtype": "f64", "frames": total_frames,
           "config": {"workload": "Iterative-F0, one %.0f s stream @%d Hz, frames of %d, time-sharded over the GPUs with a "
                                  "%d-sample halo, one all_gather of [frames, 12] (BASELINE.json configs[4]); the stream "
                                  "is resident in HBM" % (secs, fs, nf_size, warm),
                      "engine_calls_per_gpu": "one call over the rank's share; the library runs it in time slices of whole "
                                              "frames under a %d GiB workspace cap (filter state carried from slice to slice)"
                                              % (stream.STREAM_WORKSPACE_BYTES >> 30)}}
    if prof:
        kms = {k: v[1] for k, v in prof.items()}
        dom = max(kms, key=kms.get)
        models = kernel_models(0, 0, 70, nf_size)
        units = {"if0_frontend_kernel": prof_samples, "if0_spectrum_kernel": -(-prof_samples // nf_size),
                 "if0_periodicity_kernel": -(-prof_samples // nf_size)}
        rec["kernels_ms"] = kms
        rec["kernels_ms_note"] = "one context, first %.0f s of this rank's shard" % (prof_samples / fs)
        rec["roofline"] = roofline_of(dom, kms[dom], units[dom], models[dom])
        rec["rooflines"] = {k: roofline_of(k, ms, units[k], models[k]) for k, ms in kms.items() if k in units}
        for k, r in list(rec["rooflines"].items()) + [(dom, rec["roofline"])]:
            with_traffic(r, "if0_stream", k)
        rec["hbm_frac_whole_path"] = (4.0 * n + 96.0 * total_frames) / wall / HBM_PEAK   # samples in once, 12 doubles per frame out (warm pass)
    if _cpu_rec(c, "if0"):
        rec["cpu_baseline"] = _cpu_rec(c, "if0")
    return rec


WORKLOADS = {"esacf_clips_4096": wl_esacf_clips, "esacf_stft_8192": wl_esacf_stft,
             "corpus_4096_all_methods": wl_corpus, "if0_stream_1h": wl_if0_stream}


if __name__ == "__main__":
    main()
