#!/bin/bash
# Prime-multiF0 per-class times of the development library under its knobs (one line per setting).
export MPX_LIB_PATH=$PWD/chord-detection_amd/libmpx_hip_dev.so MPX_PRIME_CLASS_MARKS=1
run() { echo "== $*"; env "$@" python3 scripts/dev/prime_time.py ${CLIPS:-4096} ${FS:-22050} 2>&1 | grep -v amdgpu.ids | tail -4; }
run MPX_NOP=1
for s in "$@"; do run $s; done
