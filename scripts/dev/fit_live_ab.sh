#!/bin/bash
# Development library only (round 6): coopfit_live_kernel (the cooperative fits running NEXT TO the lane kernel, early hand-off of
# the fits that look like runaways) against the serial arrangement, over scripts/dev/esacf_time.py's shapes.
#   bash scripts/dev/fit_live_ab.sh   (on the GPU box, from the repo root)
export MPX_LIB_PATH=$PWD/chord-detection_amd/libmpx_hip_dev.so
run() { echo "== $*"; env "$@" timeout 200 python3 scripts/dev/esacf_time.py 2>&1 | grep -v amdgpu; env "$@" MPX_DEBUG_FITS=1 ESACF_TIME_ONLY=stft timeout 100 python3 scripts/dev/esacf_time.py 2>&1 | grep "^mpx esacf" | tail -1; }
run MPX_FIT_LIVE=0
run MPX_NOP=1
run MPX_FIT_EARLY_NFEV=12
run MPX_FIT_EARLY_NFEV=40
run MPX_FIT_PARK_NFEV=100
run MPX_FIT_EARLY_CAP=6000
