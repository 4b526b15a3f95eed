#!/bin/bash
# Development library only (round 6): coopfit_live_kernel (the cooperative fits running NEXT TO the lane kernel, early hand-off of
# the fits that look like runaways) against the serial arrangement, on the 8192-frame Target (scripts/dev/esacf_time.py).
#   bash scripts/dev/fit_live_ab.sh   (on the GPU box, from the repo root)
export MPX_LIB_PATH=$PWD/chord-detection_amd/libmpx_hip_dev.so ESACF_TIME_ONLY=stft
run() { echo "== $*"; env "$@" timeout 100 python3 scripts/dev/esacf_time.py 2>&1 | grep -v amdgpu; env "$@" MPX_DEBUG_FITS=1 timeout 100 python3 scripts/dev/esacf_time.py 2>&1 | grep "^mpx esacf" | tail -1; }
run MPX_FIT_LIVE=0
run MPX_FIT_LIVE=1
run MPX_FIT_LIVE=1 MPX_FIT_LIVE_POLL=0
run MPX_FIT_LIVE=1 MPX_FIT_LIVE_POLL=31
run MPX_FIT_LIVE=1 MPX_FIT_LIVE_HALF=0 MPX_FIT_EARLY_NFEV=0
run MPX_FIT_LIVE=1 MPX_FIT_LIVE_HALF=0 MPX_FIT_EARLY_NFEV=0 MPX_FIT_PARK_NFEV=100
run MPX_FIT_LIVE=1 MPX_FIT_LIVE_HALF=0 MPX_FIT_EARLY_NFEV=0 MPX_FIT_PARK_NFEV=60
run MPX_FIT_LIVE=1 MPX_FIT_LIVE_HALF=0 MPX_FIT_EARLY_NFEV=20
run MPX_FIT_LIVE=1 MPX_FIT_LIVE_HALF=0 MPX_FIT_EARLY_NFEV=40 MPX_FIT_EARLY_CAP=8000
run MPX_FIT_LIVE=1 MPX_FIT_BLOCKS=768 MPX_FIT_LIVE_HALF=0 MPX_FIT_EARLY_NFEV=20
