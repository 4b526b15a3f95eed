"""ONE trial of the bench's corpus sequence in a fresh process: a 1024-clip pass (workspaces of that size), the streaming
run, then the timed 4096-clip group -- whose larger workspaces are allocated INSIDE the timed call, on three contexts at once.
Prints the timed wall and when each method's call returned."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from chord_detection_amd import corpus
fs, n, dev = 22050, 4096, "cuda:0"
corpus.SIDE_THREADS_WAIT = (sys.argv[1:] or ["wait"])[0] == "wait"   # "nowait": the side threads start at once (rounds 3-5)
corpus.run_corpus(1024, (1, 2, 3, 4), fs, 2.0, 1024, 0, 1, 0, synth_device=dev)
block = corpus.synth_block(n, fs, 2.0, 1024, 0, 1, synth_device=dev)
corpus.run_corpus(n, (1, 2, 3, 4), fs, 2.0, 1024, 0, 1, 0, synth_device=dev)
torch.cuda.synchronize()
print("---- the timed group", file=sys.stderr, flush=True)
t0 = time.perf_counter()
lo, hi, out, spent = corpus.run_corpus(n, (1, 2, 3, 4), fs, 2.0, 1024, 0, 1, 0, synth_device=dev, resident=block)
wall = time.perf_counter() - t0
print("gate: waited %.2f ms in %d polls" % (1e3 * corpus.LAST_GATE_SECONDS[0], corpus.LAST_GATE_SECONDS[1]), file=sys.stderr, flush=True)
print("---- the same group again", file=sys.stderr, flush=True)
t0 = time.perf_counter()
corpus.run_corpus(n, (1, 2, 3, 4), fs, 2.0, 1024, 0, 1, 0, synth_device=dev, resident=block)
again = time.perf_counter() - t0
print("side threads %s: timed group %.1f ms (methods 1-4 returned after %s ms); the same group again %.1f ms" % (
    "wait for the main context's first work" if corpus.SIDE_THREADS_WAIT else "start at once", 1e3 * wall, " ".join("%.1f" % (1e3 * v) for v in spent), 1e3 * again), flush=True)
