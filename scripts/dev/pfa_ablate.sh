#!/bin/bash
# sacf_pfa_kernel by phase (development library: MPX_SACF_ABLATE bits -- 1 powers, 2 peak picking, 4 second transform, 8 first
# transform; results are garbage, times are what is read): the clip batches of scripts/dev/esacf_time.py
export MPX_LIB_PATH=$PWD/chord-detection_amd/libmpx_hip_dev.so ESACF_TIME_ONLY=frame
for ab in 0 1 2 4 8 12 15; do echo "== MPX_SACF_ABLATE=$ab"; MPX_SACF_ABLATE=$ab timeout 200 python3 scripts/dev/esacf_time.py 2>&1 | grep -v amdgpu | sed -e "s/'peakfit.*//" | cut -c1-150; done
