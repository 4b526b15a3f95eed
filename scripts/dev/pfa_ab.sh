R=$PWD
for rep in 1 2; do
for lib in enhbase dev nokeep1; do
  echo "== $lib"; MPX_LIB_PATH=$R/chord-detection_amd/libmpx_hip_$lib.so timeout 200 python3 scripts/dev/esacf_time.py 2>&1 | grep -v amdgpu | grep clips
done; done
