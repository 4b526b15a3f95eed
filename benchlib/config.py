"""Sizes and peaks of the benchmark (BASELINE.json configs[1..4]).  Module attributes on purpose: the CPU
tests' stand-in backend (tests/bench_stub.py) shrinks them before anything runs, so every user reads them at call time."""
FS, N_FFT, HOP, FRAMES = 44100, 4096, 1024, 8192
F_ALG = int(2.5 * 4096 * 12 + 4096)  # 2.5 N log2 N + N at N = 4096
B_ALG = 4 * HOP + 48          # compulsory HBM bytes per frame, overlapped-signal input
PREHEAT_MS = 100              # untimed launches before the W warm-up steps: clock ramp of a cold device (bench.py main)
HBM_PEAK = 8.0e12             # MI355X_MICROARCH.md: 8.0 TB/s spec
F64_PEAK = 78.65e12           # fp64 vector: half the 157.3 TFLOP/s FP32 vector rate (same guide)
NSIG = 9                      # distinct input signals the steps rotate over: 9 x 33.5 MB = 302 MB > the 256 MiB MALL
# sizes of the secondary workloads (a stand-in backend for the CPU tests shrinks them)
CFG = {"esacf_clips": 4096, "esacf_fs": 44100, "esacf_clip_seconds": 2.0, "corpus_clips_per_gpu": 4096,
       "corpus_fs": 22050, "stream_seconds": 3600.0, "stream_fs": 44100, "if0_frame": 8192,
       "he_default_clips": 1366, "he_default_fs": 22050, "he_default_frame": 8192}
