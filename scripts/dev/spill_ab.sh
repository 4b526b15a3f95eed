#!/bin/bash
# the kernels that are ALLOWED to spill (tests/test_kernel_resources.py SCRATCH_CEILING) because the spill buys a
# resident workgroup more per CU -- the trade, re-measurable on any box: the shipping build against a library whose kernel
# is held to one workgroup fewer (scratch-free).  Build step HERE (no GPU needed), timing step on the GPU box:
#   bash scripts/dev/spill_ab.sh build      # -> chord-detection_amd/libmpx_hip_per3.so, libmpx_hip_pv2.so (development builds)
#   bash scripts/dev/spill_ab.sh run        # on the GPU box, from the repo root
cd "$(dirname "$0")/../.."
C=chord-detection_amd/csrc; F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-value -Wno-unused-result -DMPX_DEV_KNOBS"
if [ "$1" = build ]; then
  make -C $C dev -j4 > /dev/null || exit 1
  /opt/rocm/bin/hipcc $F -DIF0_PER_WGS=3 -c $C/mpx_if0.hip -o /tmp/per3_if0.o && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $C/dev_mpx_api.o $C/dev_mpx_he.o $C/dev_mpx_esacf.o $C/dev_mpx_prime.o /tmp/per3_if0.o -o chord-detection_amd/libmpx_hip_per3.so
  /opt/rocm/bin/hipcc $F -DPV_WGS=2 -c $C/mpx_esacf.hip -o /tmp/pv2_esacf.o && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $C/dev_mpx_api.o $C/dev_mpx_he.o /tmp/pv2_esacf.o $C/dev_mpx_prime.o $C/dev_mpx_if0.o -o chord-detection_amd/libmpx_hip_pv2.so
  ls -la chord-detection_amd/libmpx_hip_per3.so chord-detection_amd/libmpx_hip_pv2.so
  exit 0
fi
R=$PWD
for rep in 1 2; do
  echo "== if0_periodicity_kernel: 4 workgroups per CU, 108 B of scratch (shipping)"; MPX_LIB_PATH=$R/chord-detection_amd/libmpx_hip_dev.so IF0_CLIPS=64 timeout 300 python3 scripts/dev/if0_time.py 2>&1 | grep -v amdgpu | head -1
  echo "== if0_periodicity_kernel: 3 workgroups per CU, scratch-free";               MPX_LIB_PATH=$R/chord-detection_amd/libmpx_hip_per3.so IF0_CLIPS=64 timeout 300 python3 scripts/dev/if0_time.py 2>&1 | grep -v amdgpu | head -1
  echo "== pv_enhance_kernel<true,2>: 3 workgroups per CU, 24 B of scratch (shipping)"; MPX_LIB_PATH=$R/chord-detection_amd/libmpx_hip_dev.so ESACF_TIME_ONLY=stft timeout 100 python3 scripts/dev/esacf_time.py 2>&1 | grep -v amdgpu
  echo "== pv_enhance_kernel<true,2>: 2 workgroups per CU, scratch-free";               MPX_LIB_PATH=$R/chord-detection_amd/libmpx_hip_pv2.so ESACF_TIME_ONLY=stft timeout 100 python3 scripts/dev/esacf_time.py 2>&1 | grep -v amdgpu
done
