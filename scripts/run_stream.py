#!/usr/bin/env python3
"""BASELINE.json configs[4]: Iterative-F0 over one long synthetic stream, time-sharded over the GPUs of one node.
1 GPU:  python scripts/run_stream.py --seconds 3600 --fs 44100
G GPUs: python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 \
            --master-port 29512 scripts/run_stream.py --seconds 3600 --fs 44100
Prints one JSON line on rank 0 (see chord-detection_amd/stream.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import chord_detection_amd  # noqa: E402,F401  (import shim for the hyphenated package directory)
from chord_detection_amd import stream  # noqa: E402

if __name__ == "__main__":
    sys.exit(stream.main())
