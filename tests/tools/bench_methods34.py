#!/usr/bin/env python3
"""Secondary benchmarks for the next-tier rows (BASELINE.json configs[3] and [4]), host-buffer entry points
(PCIe copy included):
  * Prime-multiF0 and Iterative-F0 on a batch of 2 s clips @22050 Hz (clips/s);
  * Iterative-F0 on one long stream @22050 Hz (x real time).
Each result is spot-checked against the oracle."""
import argparse, json, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import chord_detection_amd as cd

FS = 22050


def clip(seed, n=44100):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / FS
    y = np.zeros(n)
    for _ in range(int(rng.integers(2, 5))):
        f0 = 440.0 * 2.0 ** ((int(rng.integers(36, 85)) - 69) / 12.0)
        for h in range(1, 9):
            if f0 * h < FS / 2:
                y += (0.7 ** (h - 1)) * np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 6))
    y += 0.003 * rng.standard_normal(n)
    return (0.9 * y / np.max(np.abs(y))).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=256)
    ap.add_argument("--stream-seconds", type=int, default=600)
    ap.add_argument("--stream-fs", type=int, default=FS, help="sample rate of the long stream (BASELINE configs[4]: 44100, 3600 s)")
    args = ap.parse_args()
    eng = cd.get_engine(0)
    uniq = [clip(20260102 + i) for i in range(16)]
    batch = [uniq[i % 16] for i in range(args.clips)]
    out = {}
    warnings.simplefilter("ignore")
    from oracle import prime_multif0 as o_prime, iterative_f0 as o_if0
    for name, fn, ofn in (("prime_multif0", eng.prime_multif0_batch, o_prime.prime_compute),
                          ("iterative_f0", eng.iterative_f0_batch, o_if0.iterative_f0_compute)):
        fn(batch[:16], FS)
        t0 = time.perf_counter()
        res = fn(batch, FS)
        dt = time.perf_counter() - t0
        ok = bool(np.allclose(res[3], ofn(batch[3], FS), rtol=1e-5, atol=1e-7))
        out[name] = {"clips": args.clips, "seconds": dt, "clips_per_s": args.clips / dt, "oracle_spot_check": ok}
    # long stream through method 3 (chunked front end)
    sfs = args.stream_fs
    n = args.stream_seconds * sfs
    reps = (n + 44099) // 44100
    stream = np.concatenate([uniq[i % 16] for i in range(reps)])[:n]   # the 22.05 kHz clips, read at the stream's rate
    eng.iterative_f0(stream[:50000], sfs)
    t0 = time.perf_counter()
    total, frames = eng.iterative_f0(stream, sfs, return_frames=True)
    dt = time.perf_counter() - t0
    want = o_if0.iterative_f0_frames(stream[:3 * 8192], sfs)[0]
    out["iterative_f0_stream"] = {"seconds_of_audio": args.stream_seconds, "fs": sfs, "seconds": dt,
                                  "x_realtime": args.stream_seconds / dt, "frames": int(frames.shape[0]),
                                  "oracle_spot_check": bool(np.allclose(frames[:3], want, rtol=1e-5))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
