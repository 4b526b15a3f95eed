#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_harmonic_energy.py -m gpu -x -q 2>&1 | tail -5
cp chord-detection_amd/libmpx_hip.so /tmp/libwave.so
timeout 300 python scripts/ab_he.py /tmp/libwave.so 2>&1 | tail -2
MPX_HE_WG=1 timeout 300 python scripts/ab_he.py /tmp/libwave.so 2>&1 | tail -2
