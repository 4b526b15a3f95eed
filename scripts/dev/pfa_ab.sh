#!/bin/bash
# same-box A/B of sacf_pfa_kernel builds (development libraries built beside libmpx_hip_dev.so: see the commit that names them)
R=$PWD
for rep in 1 2; do
for lib in pick4 dev; do
  echo "== $lib"; MPX_LIB_PATH=$R/chord-detection_amd/libmpx_hip_$lib.so ESACF_TIME_ONLY=frame timeout 200 python3 scripts/dev/esacf_time.py 2>&1 | grep -v amdgpu | grep clips
done; done
