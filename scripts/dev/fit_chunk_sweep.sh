#!/bin/bash
# round 6: items a wave of the lane fit kernel reserves per returning atomic (development library; 0 = what it needs)
export MPX_LIB_PATH=$PWD/chord-detection_amd/libmpx_hip_dev.so
for ch in 0 32 0 8 16 32 64; do echo "== MPX_FIT_CHUNK=$ch"; MPX_FIT_CHUNK=$ch timeout 200 python3 scripts/dev/esacf_time.py 2>&1 | grep -v amdgpu | cut -c1-235; done
