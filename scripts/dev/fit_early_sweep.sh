#!/bin/bash
# Development library only (round 6): the early-parking rule of peakfit_kernel (fits that LOOK like runaways leave the lane kernel
# at once) against the build without it, and the end-game threshold next to it, over scripts/dev/esacf_time.py's three shapes.
#   bash scripts/dev/fit_early_sweep.sh   (on the GPU box, from the repo root)
export MPX_LIB_PATH=$PWD/chord-detection_amd/libmpx_hip_dev.so
run() { echo "== $*"; env "$@" MPX_DEBUG_FITS=1 timeout 200 python3 scripts/dev/esacf_time.py 2>&1 | grep -v amdgpu | grep -v "^mpx esacf" ; env "$@" MPX_DEBUG_FITS=1 ESACF_TIME_ONLY=stft timeout 100 python3 scripts/dev/esacf_time.py 2>&1 | grep "^mpx esacf" | tail -1; }
run MPX_FIT_EARLY_NFEV=0
run MPX_NOP=1
run MPX_FIT_PARK_NFEV=100
run MPX_FIT_PARK_NFEV=60
run MPX_FIT_EARLY_NFEV=12
run MPX_FIT_EARLY_NFEV=40
run MPX_FIT_EARLY_NFEV=20 MPX_COOP_PASS1_TRIPS=12
run MPX_FIT_EARLY_NFEV=20 MPX_FIT_PARK_NFEV=100 MPX_COOP_PASS1_TRIPS=40
