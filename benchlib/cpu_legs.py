"""CPU legs: the oracle (a NumPy port of the reference's math) on the host cores of this box.  Reported, not a target.
Run FIRST, before anything touches the GPU: they fork worker processes."""
import os
import time

from . import config as K
from .synth import synth_signal, synth_clips_numpy


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
    nfr = (x.shape[0] - K.N_FFT) // K.HOP + 1
    chunk, done, t0, f = 128, 0, time.perf_counter(), worker * 7
    while time.perf_counter() - t0 < budget_s:
        lo = (f % (nfr // chunk)) * chunk * K.HOP
        o_he.he_frames(x[lo:lo + (chunk - 1) * K.HOP + K.N_FFT], K.FS, K.N_FFT, K.HOP)
        done += chunk
        f += 1
    return done, time.perf_counter() - t0


def _cpu_he_default(budget_s, worker):
    """Harmonic Energy at the reference's default shape: 8192-sample frames, hop = frame, 22.05 kHz clips, 6 frames per call."""
    from oracle import harmonic_energy as o_he
    clips, fs, nfr = _CPU_INPUT["clips22x6"], K.CFG["he_default_fs"], K.CFG["he_default_frame"]
    done, t0, f = 0, time.perf_counter(), worker
    while time.perf_counter() - t0 < budget_s:
        o_he.he_frames(clips[f % clips.shape[0]], fs, nfr)
        done += clips.shape[1] // nfr
        f += 1
    return done, time.perf_counter() - t0


def _cpu_esacf(budget_s, worker, frame, hop, fs):
    import warnings
    from oracle import esacf as o_esacf
    x = _CPU_INPUT["he"] if fs == K.FS and hop != frame else _CPU_INPUT["clip44"]
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
    fs = K.CFG["esacf_fs"]
    return _cpu_esacf(budget_s, worker, int(fs * 46.4 / 1000), int(fs * 46.4 / 1000), fs)


def _cpu_esacf_stft(budget_s, worker):
    return _cpu_esacf(budget_s, worker, K.N_FFT, K.HOP, K.FS)


def _cpu_corpus(budget_s, worker):
    """All four methods on whole 2 s clips @22.05 kHz, one clip after the other like the reference's tests/test.py loop."""
    import warnings
    from oracle import esacf as o_esacf, harmonic_energy as o_he, iterative_f0 as o_if0, prime_multif0 as o_prime
    clips, fs = _CPU_INPUT["clips22"], K.CFG["corpus_fs"]
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
    x, fs, nf = _CPU_INPUT["he"], K.CFG["stream_fs"], K.CFG["if0_frame"]
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
             "if0": _cpu_if0, "he_default": _cpu_he_default}


def _cpu_worker(arg):
    name, budget, worker = arg
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)          # one thread per worker
    except Exception:
        pass
    return _CPU_LEGS[name](budget, worker)


def cpu_baselines(budget_s=6.0, legs=("he", "esacf_clips", "esacf_stft", "corpus", "if0", "he_default")):
    """Every leg twice: one process on one core, then one process per physical core (multiprocessing, fork): units
    done / wall clock of the slowest worker.  Bounded samples of the same workloads the GPU legs run."""
    import multiprocessing as mp
    info = host_cpu_info()
    _CPU_INPUT["he"] = synth_signal(20260101, frames=2048)
    _CPU_INPUT["clips22"] = synth_clips_numpy(16, K.CFG["corpus_fs"], 2.0)
    _CPU_INPUT["clip44"] = synth_clips_numpy(4, K.CFG["esacf_fs"], 2.0).reshape(-1)
    _CPU_INPUT["clips22x6"] = synth_clips_numpy(8, K.CFG["he_default_fs"], 6.0 * K.CFG["he_default_frame"] / K.CFG["he_default_fs"])
    from oracle import esacf, harmonic_energy, iterative_f0, prime_multif0  # noqa: F401  (imported before the fork)
    units = {"he": "frames/s", "esacf_clips": "frames/s", "esacf_stft": "frames/s", "corpus": "clips/s",
             "if0": "x real time", "he_default": "frames/s"}
    samples = {
        "he": "N=4096 hop=1024 frames of a 47 s stretch of the bench signal through oracle/harmonic_energy.py (numpy.fft.rfft, float64, 128 frames per call)",
        "esacf_clips": "46.4 ms frames of 44.1 kHz polyphonic clips through oracle/esacf.py (4 frames per call)",
        "esacf_stft": "N=4096 hop=1024 frames of the bench signal through oracle/esacf.py (phase-vocoder regime, 4 frames per call)",
        "corpus": "2 s clips @22.05 kHz through all four oracle methods, one clip after the other",
        "if0": "32768-sample pieces of a 44.1 kHz signal through oracle/iterative_f0.py",
        "he_default": "clips of six 8192-sample frames @22.05 kHz through oracle/harmonic_energy.py (hop = frame, one clip per call)",
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
        o_he.he_frames(x[i * K.HOP:i * K.HOP + K.N_FFT], K.FS, K.N_FFT)
    out["he"]["per_frame_loop_frames_per_s"] = k / (time.perf_counter() - t1)
    for rec in out.values():
        rec["host"] = info
    import numpy
    out["he"]["numpy"] = numpy.__version__
    return out

