#!/usr/bin/env python3
"""TESTS ONLY: run one of the multi-GPU driver SCRIPTS (scripts/run_corpus.py, scripts/run_stream.py) as a rank of a
CPU job: the script itself is executed (runpy), with the driver's injection points set -- a checker instead of the
engine, CPU tensors, gloo -- so that everything else (partitioning, halos, the one all_gather, max-over-ranks timing,
rank 0's JSON line) is the code an 8-GPU run executes.  Usage: launch_cpu_rank.py corpus|stream [driver args...]"""
import functools
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import chord_detection_amd  # noqa: E402,F401
from chord_detection_amd import corpus, stream  # noqa: E402
from tests import bench_stub  # noqa: E402

which, rest = sys.argv[1], sys.argv[2:]
if which == "corpus":
    corpus.main = functools.partial(corpus.main, compute=bench_stub.corpus_compute, device="cpu", backend="gloo")
    script = "run_corpus.py"
else:
    stream.main = functools.partial(stream.main, compute=bench_stub.stream_compute, device="cpu", backend="gloo")
    script = "run_stream.py"
sys.argv = [os.path.join(ROOT, "scripts", script)] + rest
runpy.run_path(sys.argv[0], run_name="__main__")
