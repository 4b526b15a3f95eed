export MPX_LIB_PATH=$PWD/chord-detection_amd/libmpx_hip_dev.so
for v in "MPX_FIT_LIVE=0" "MPX_FIT_LIVE=1" "MPX_FIT_LIVE=1 MPX_FIT_PARK_NFEV=100" "MPX_FIT_LIVE=0" "MPX_FIT_LIVE=1" "MPX_FIT_LIVE=1 MPX_FIT_PARK_LIVE=16 MPX_FIT_PARK_CAP=65536"; do echo "== $v"; env $v timeout 200 python3 scripts/dev/esacf_time.py 2>&1 | grep -v amdgpu; done
