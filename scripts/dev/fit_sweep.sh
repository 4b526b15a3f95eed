#!/bin/bash
# Development library only: sweeps of the fit scheduling knobs over scripts/dev/esacf_time.py's three shapes
# (round 5: resident blocks of the lane kernel, the parking threshold; profiles/r5/fit_sweep_*.txt).
#   bash scripts/dev/fit_sweep.sh   (on the GPU box, from the repo root)
export MPX_LIB_PATH=$PWD/chord-detection_amd/libmpx_hip_dev.so
run() { echo "== $*"; env "$@" timeout 200 python3 scripts/dev/esacf_time.py 2>&1 | grep -v amdgpu; }
run MPX_NOP=1
for b in 256 512 768; do run MPX_FIT_BLOCKS=$b; done
for nf in 60 100 220; do run MPX_FIT_PARK_NFEV=$nf; done
