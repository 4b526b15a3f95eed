#!/bin/bash
# what the two returning atomics per frame on the work list's counters cost sacf_pfa_kernel (development ablation, results garbage)
export MPX_LIB_PATH=$PWD/chord-detection_amd/libmpx_hip_dev.so ESACF_TIME_ONLY=frame
for rep in 1 2; do for ab in 0 32; do echo "== MPX_SACF_ABLATE=$ab"; MPX_SACF_ABLATE=$ab timeout 200 python3 scripts/dev/esacf_time.py 2>&1 | grep -v amdgpu | sed -e "s/'peakfit.*//" | cut -c1-150; done; done
