R=$PWD
timeout 1300 python3 -m pytest tests/test_gpu_esacf.py tests/test_gpu_zz_fragile_table.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2 3; do MPX_LIB_PATH=$R/chord-detection_amd/libmpx_hip_dev.so timeout 200 python3 scripts/dev/esacf_time.py 2>&1 | grep -v amdgpu; done
FUZZ_SIZES=1023,2046,4096,700 timeout 400 python3 tests/tools/fuzz_esacf.py 200 691 2>&1 | grep -v amdgpu | tail -1
FUZZ_WIDE=1 timeout 400 python3 tests/tools/fuzz_esacf.py 200 692 2>&1 | grep -v amdgpu | tail -1
