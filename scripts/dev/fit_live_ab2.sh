#!/bin/bash
# round 6: the end-game threshold with coopfit_live_kernel taking the tail (full lane grid, no early hand-off)
export MPX_LIB_PATH=$PWD/chord-detection_amd/libmpx_hip_dev.so ESACF_TIME_ONLY=stft
run() { echo "== $*"; env "$@" timeout 100 python3 scripts/dev/esacf_time.py 2>&1 | grep -v amdgpu; }
run MPX_FIT_LIVE=0
run MPX_NOP=1
run MPX_FIT_PARK_NFEV=130
run MPX_FIT_PARK_NFEV=200
run MPX_FIT_PARK_NFEV=240
run MPX_FIT_LIVE_POLL=0
run MPX_FIT_LIVE_POLL=3
run MPX_FIT_PARK_LIVE=16
run MPX_FIT_PARK_LIVE=64 MPX_FIT_PARK_CAP=100000
run MPX_NOP=2
unset ESACF_TIME_ONLY
run MPX_NOP=3
