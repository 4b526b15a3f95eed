#!/usr/bin/env python3
"""BASELINE.json configs[3]: all four methods over a synthetic corpus, clip-sharded over the GPUs of one node.
1 GPU:  python scripts/run_corpus.py --clips 4096
G GPUs: python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 \
            --master-port 29511 scripts/run_corpus.py --clips 100000
Prints one JSON line on rank 0 (see chord-detection_amd/corpus.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import chord_detection_amd  # noqa: E402,F401  (import shim for the hyphenated package directory)
from chord_detection_amd import corpus  # noqa: E402

if __name__ == "__main__":
    sys.exit(corpus.main())
