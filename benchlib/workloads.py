"""Secondary workloads of bench.py: each returns a record on rank 0 (None elsewhere)."""
import math
import sys
import time

from . import config as K
from .roofline import kernel_models, roofline_of, fit_roofline, dominant, with_traffic


def _max_over_ranks(c, seconds):
    t = c["torch"].tensor([seconds], dtype=c["torch"].float64, device=c["dev"])
    if c.get("use_dist", c["world"] > 1):
        c["dist"].all_reduce(t, op=c["dist"].ReduceOp.MAX)
    return float(t.item())


def _cpu_rec(c, leg):
    return c["cpu"][leg] if c["cpu"] is not None else None


def wl_esacf_clips(c):
    """configs[2]: every rank runs its own 4096 clips (weak scaling) through the batch entry point of the C ABI, the clips
    resident in HBM; per-clip framing (44 frames of 2046 samples per 2 s clip @44.1 kHz)."""
    torch, np, eng = c["torch"], c["np"], c["eng"]
    from chord_detection_amd import corpus
    fs, secs, clips = K.CFG["esacf_fs"], K.CFG["esacf_clip_seconds"], K.CFG["esacf_clips"]
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
    rec["hbm_frac_whole_path"] = (4 * frame + 48) * rec["value"] / c["world"] / K.HBM_PEAK
    if _cpu_rec(c, "esacf_clips"):
        rec["cpu_baseline"] = _cpu_rec(c, "esacf_clips")
    return rec


def wl_esacf_stft(c):
    """The north star's Target: STFT -> ESACF -> chromagram on ONE signal of 8192 overlapping frames (N=4096, hop 1024),
    device-resident (mpx_esacf_dev); one launch sequence at a time, and with three batches in flight on three contexts."""
    torch, np, eng = c["torch"], c["np"], c["eng"]
    sigs, n = c["sigs"], c["sigs"][0].numel()
    nf = K.FRAMES
    outs = [(torch.zeros((nf, 12), dtype=torch.float64, device=c["dev"]), torch.zeros(12, dtype=torch.float64, device=c["dev"]))
            for _ in range(3)]
    engs = (c["engs"] + [c["make_engine"]() for _ in range(3)])[:3]
    for e, (fr, sm) in zip(engs, outs):
        e.esacf_dev(sigs[0].data_ptr(), n, K.FS, K.N_FFT, K.HOP, fr.data_ptr(), sm.data_ptr())
        e.synchronize()
    reps = 6
    c["barrier"]()
    t0 = time.perf_counter()
    for r in range(reps):
        eng.esacf_dev(sigs[r % len(sigs)].data_ptr(), n, K.FS, K.N_FFT, K.HOP, outs[0][0].data_ptr(), outs[0][1].data_ptr())
    eng.synchronize()
    c["barrier"]()
    wall1 = _max_over_ranks(c, (time.perf_counter() - t0) / reps)
    c["barrier"]()
    t0 = time.perf_counter()
    for r in range(3 * reps):
        e, (fr, sm) = engs[r % 3], outs[r % 3]
        e.esacf_dev(sigs[r % len(sigs)].data_ptr(), n, K.FS, K.N_FFT, K.HOP, fr.data_ptr(), sm.data_ptr())
    for e in engs:
        e.synchronize()
    c["barrier"]()
    wall3 = _max_over_ranks(c, (time.perf_counter() - t0) / (3 * reps))
    eng.profile_begin()
    eng.esacf_dev(sigs[0].data_ptr(), n, K.FS, K.N_FFT, K.HOP, outs[0][0].data_ptr(), outs[0][1].data_ptr())
    eng.synchronize()
    prof = eng.profile_end()
    stats = eng.esacf_fit_stats() if hasattr(eng, "esacf_fit_stats") else None
    ok = None
    if c["stub"] is None:
        import warnings
        from oracle import esacf as o_esacf
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = o_esacf.esacf_frames(c["x_host"][:2 * K.HOP + K.N_FFT].astype(np.float64), K.FS, K.N_FFT, K.HOP)
        ok = bool(np.allclose(outs[0][0][:3].cpu().numpy(), want[:3], rtol=1e-5, atol=1e-12))
    if c["rank"] != 0:
        return None
    kms = {k: v[1] / max(v[0], 1) for k, v in prof.items()}
    models = kernel_models(K.N_FFT, (K.N_FFT - 1) // 2)
    dom, dms = dominant(prof)
    rec = {"value": nf * c["world"] / wall1, "unit": "frames/s", "ms_per_batch": 1e3 * wall1,
           "value_one_call": nf * c["world"] / wall1, "ms_one_call": 1e3 * wall1,   # = value: one call after the other
           "value_three_in_flight": nf * c["world"] / wall3, "ms_per_batch_three_in_flight": 1e3 * wall3,
           "scaling": "weak", "dtype": "f64",
           "config": {"workload": "STFT->ESACF->chromagram, one signal of 8192 frames per GPU, N=4096 hop 1024 @44.1 kHz "
                                  "(BASELINE.json north_star Target)", "frames_per_gpu": nf, "entry": "mpx_esacf_dev"},
           "kernels_ms": kms, "kernel_ms_total": sum(kms.values()), "oracle_spot_check": ok,
           "roofline": roofline_of(dom, dms, nf, models[dom]) if dom in models else
           (fit_roofline(kms, stats) if stats else {"kernel": dom, "kernel_ms": dms, "bound": "latency", "frac": None, "traffic": None}),
           "rooflines": {k: roofline_of(k, ms, nf, models[k]) for k, ms in kms.items() if k in models},
           "hbm_frac_whole_path": K.B_ALG * nf / wall1 / K.HBM_PEAK,
           "hbm_frac_whole_path_three_in_flight": K.B_ALG * nf / wall3 / K.HBM_PEAK}
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
    per, world, rank = K.CFG["corpus_clips_per_gpu"], c["world"], c["rank"]
    fs = K.CFG["corpus_fs"]
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
    ud = c.get("use_dist", world > 1)
    chroma = corpus.gather_blocks(block, per * world, world, rank, c["dev"] if ud and c["stub"] is None else None, force=ud)
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
    fs, secs, nf_size = K.CFG["stream_fs"], K.CFG["stream_seconds"], K.CFG["if0_frame"]
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

    # Both timed passes run WITH the library's kernel profile on (two event records per launch, 42 over the hour): the
    # record's kernel times and roofline fractions are those of the timed calls themselves, not of a side run.
    prof_eng = c["cd"].get_engine(local) if c["stub"] is None else None
    if prof_eng is not None:
        prof_eng.profile_begin()
    t0 = time.perf_counter()
    block = compute_block()           # first pass: grows the contexts' workspaces (tens of GB of hipMalloc)
    cold = time.perf_counter() - t0
    prof = prof_eng.profile_end() if prof_eng is not None else {}
    c["barrier"]()
    if prof_eng is not None:
        prof_eng.profile_begin()
    t0 = time.perf_counter()
    block = compute_block()
    t_warm_engine = time.perf_counter() - t0
    prof_warm = prof_eng.profile_end() if prof_eng is not None else {}
    ud = c.get("use_dist", world > 1)
    frames = stream.gather_frames(block, total_frames, world, rank, c["dev"] if ud and c["stub"] is None else None, force=ud)
    c["barrier"]()
    wall = _max_over_ranks(c, time.perf_counter() - t0)
    cold = _max_over_ranks(c, cold)
    prof_samples = int(x.numel())   # this rank's share of the stream, run-in halo included
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
        rec["kernels_ms_warm_pass"] = {k: v[1] for k, v in prof_warm.items()}
        rec["kernels_ms_note"] = ("HIP-event times of the kernels INSIDE the timed first pass (`value`), this rank's whole share "
                                  "(%.0f s of audio, %d time slices' launches summed); kernels_ms_warm_pass: the same inside "
                                  "the timed second pass (`value_warm`)" % (prof_samples / fs, max(v[0] for v in prof.values())))
        rec["warm_pass_engine_call_s"] = t_warm_engine
        rec["roofline"] = roofline_of(dom, kms[dom], units[dom], models[dom])
        rec["rooflines"] = {k: roofline_of(k, ms, units[k], models[k]) for k, ms in kms.items() if k in units}
        for k, r in list(rec["rooflines"].items()) + [(dom, rec["roofline"])]:
            # (the PMC passes ran a 600 s stream, scripts/pmc_workloads.py: scaled to this rank's share)
            with_traffic(r, "if0_stream", k, launches=prof_samples / (600.0 * fs))
        rec["hbm_frac_whole_path"] = (4.0 * n + 96.0 * total_frames) / wall / K.HBM_PEAK   # samples in once, 12 doubles per frame out (warm pass)
        # What the three kernels move through HBM together (the last PMC collection, scaled to this rank's share) against the
        # bytes of the samples themselves: the fp64 70-channel hand-off between front end and spectra dominates it -- the
        # per-kernel `traffic_vs_compulsory_plus_intermediate` figures (~1.0-1.2: no re-reads) do not show that.
        tr = [r.get("traffic") for r in rec["rooflines"].values()]
        if tr and all(t is not None for t in tr):
            rec["traffic_whole_path_bytes"] = float(sum(tr))
            rec["traffic_vs_sample_bytes"] = float(sum(tr)) / (4.0 * prof_samples)
    if _cpu_rec(c, "if0"):
        rec["cpu_baseline"] = _cpu_rec(c, "if0")
    return rec


def wl_he_default(c):
    """The reference's own default Harmonic-Energy shape (harmonic_energy.py:14-16: frame_size 8192, non-overlapping frames,
    librosa's 22.05 kHz): a batch of clips of six whole frames each through the batch entry point, clips resident in HBM.
    Independent frames: SURVEY 8(d) roofline (ii), B_alg = 4 N + 48 bytes per frame."""
    torch, np, eng = c["torch"], c["np"], c["eng"]
    from chord_detection_amd import corpus
    fs, clips, frame = K.CFG["he_default_fs"], K.CFG["he_default_clips"], K.CFG["he_default_frame"]
    per_clip = 6
    secs = per_clip * frame / fs
    uniq = corpus.synth_chunk(list(range(64 * c["rank"], 64 * c["rank"] + min(64, clips))), fs, secs, c["dev"])
    x = uniq.repeat((clips + uniq.shape[0] - 1) // uniq.shape[0], 1)[:clips, :per_clip * frame].contiguous()
    assert x.shape[1] == per_clip * frame
    frames = clips * per_clip
    flat = x.reshape(-1)
    flats = [flat, torch.roll(flat, 12345)]   # two 268 MB inputs in turn: more than the 256 MiB Infinity Cache holds
    d_rows = torch.zeros((frames, 12), dtype=torch.float64, device=c["dev"])
    first = eng.harmonic_energy_batch(x, fs, frame)          # plans, workspaces
    reps = 20
    c["barrier"]()
    t0 = time.perf_counter()
    for _ in range(reps):
        got = eng.harmonic_energy_batch(x, fs, frame)       # per-clip 12-vectors back on the host: the drop-in's batch call
    c["barrier"]()
    wall = _max_over_ranks(c, (time.perf_counter() - t0) / reps)
    # the kernel alone: the same frames as ONE signal with hop = frame (the frames of the clips back to back), rows out
    kreps = 50
    for _ in range(3):
        eng.harmonic_energy_dev(flat.data_ptr(), flat.numel(), fs, frame, frame, d_rows.data_ptr(), None)
    eng.synchronize()
    eng.timer_begin()
    for r in range(kreps):
        eng.harmonic_energy_dev(flats[r & 1].data_ptr(), flat.numel(), fs, frame, frame, d_rows.data_ptr(), None)
    kern_ms = eng.timer_end() / kreps
    eng.synchronize()
    eng.harmonic_energy_dev(flat.data_ptr(), flat.numel(), fs, frame, frame, d_rows.data_ptr(), None)   # (the rows checked below)
    eng.synchronize()
    if not np.array_equal(got, first):
        sys.exit("bench: Harmonic-Energy batch results differ between runs (non-deterministic)")
    ok = None
    if c["stub"] is None:   # spot check against the oracle: first clip whole, and the kernel-only rows of its frames
        from oracle import harmonic_energy as o_he
        x0 = x[0].cpu().numpy()
        want = o_he.he_frames(x0, fs, frame)
        ok = bool(np.allclose(got[0], want.sum(0), rtol=1e-9) and np.allclose(d_rows[:per_clip].cpu().numpy(), want, rtol=1e-9))
    if c["rank"] != 0:
        return None
    b_alg = 4 * frame + 48
    f_alg = 2.5 * frame * math.log2(frame) + frame
    hbm = b_alg * frames / (kern_ms * 1e-3)
    fl = f_alg * frames / (kern_ms * 1e-3)
    rec = {"value": frames * c["world"] / wall, "unit": "frames/s", "ms_per_batch": 1e3 * wall, "scaling": "weak", "dtype": "f64",
           "value_kernel_only": frames / (kern_ms * 1e-3),
           "config": {"workload": "Harmonic Energy at the reference's default shape: %d clips x %d frames of %d samples, hop = frame, "
                                  "@%d Hz per GPU (harmonic_energy.py:14-16)" % (clips, per_clip, frame, fs),
                      "frames_per_gpu": frames, "entry": "mpx_harmonic_energy_batch, clips in HBM; kernel_ms: mpx_harmonic_energy_dev"},
           "kernels_ms": {"he_kernel": kern_ms}, "oracle_spot_check": ok,
           "roofline": {"bound": "hbm", "achieved": hbm / 1e9, "peak": K.HBM_PEAK / 1e9, "unit": "GB/s", "frac": hbm / K.HBM_PEAK,
                        "kernel": "he_wave_kernel<6,4,...,2,pairs> (a pair of waves per 8192-sample frame: a streamed call, round 6)", "kernel_ms": kern_ms, "units_per_launch": frames, "unit_of_work": "frame",
                        "bytes_per_unit": b_alg, "flops_per_unit": f_alg, "hbm_frac": hbm / K.HBM_PEAK,
                        "valu_f64_frac": fl / K.F64_PEAK, "compulsory_bytes": b_alg * frames, "traffic": None},
           "hbm_frac_whole_path": b_alg * frames / wall / K.HBM_PEAK}
    with_traffic(rec["roofline"], "he_default", "he_kernel")
    if _cpu_rec(c, "he_default"):
        rec["cpu_baseline"] = _cpu_rec(c, "he_default")
    return rec


WORKLOADS = {"esacf_clips_4096": wl_esacf_clips, "esacf_stft_8192": wl_esacf_stft,
             "corpus_4096_all_methods": wl_corpus, "if0_stream_1h": wl_if0_stream, "he_default_8192": wl_he_default}
