#!/bin/bash
mkdir -p gpurun_out/r2w
timeout 600 python -m pytest tests/test_gpu_harmonic_energy.py tests/test_gpu_bench_contract.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r2w/gputest_he.txt
timeout 1500 python3 bench.py --steps 2000 --warmup 200 > gpurun_out/r2w/bench_plain.json 2> gpurun_out/r2w/bench_plain.err
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/kp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2000 --warmup 200 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2w/bench_profiled.json 2> /tmp/kp.err
cp $(find /tmp/kp -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r2w/bench_kernel_stats.csv
rm -rf /tmp/kp1; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2000 --warmup 200 --streams 1 --no-cpu-baseline --headline-only > $GRAFT_REPO_ROOT/gpurun_out/r2w/bench_streams1_profiled.json 2> /tmp/kp1.err
cp $(find /tmp/kp1 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r2w/bench_streams1_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tr_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/tr_$C -- python3 $GRAFT_REPO_ROOT/scripts/traffic_probe.py > /tmp/tr_$C.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 scripts/traffic_summary.py $(find /tmp/tr_FETCH_SIZE -name "*counter_collection.csv") $(find /tmp/tr_WRITE_SIZE -name "*counter_collection.csv") > gpurun_out/r2w/traffic_he_wave.txt 2>&1
cp gpurun_out/traffic_latest.json gpurun_out/r2w/
cat gpurun_out/r2w/gputest_he.txt; cut -c1-900 gpurun_out/r2w/bench_plain.json; tail -2 gpurun_out/r2w/bench_plain.err; head -5 gpurun_out/r2w/bench_kernel_stats.csv; head -4 gpurun_out/r2w/bench_streams1_kernel_stats.csv; cat gpurun_out/r2w/traffic_he_wave.txt
(timeout 60 ./build_tmp/he_wave_check 8192; timeout 60 ./build_tmp/he_wave_check 32768 | tail -3) > gpurun_out/r2w/he_wave_check.txt 2>&1
bash scripts/dev/pmc_he_wave.sh > gpurun_out/r2w/pmc_he_wave.txt 2>&1
timeout 60 ./build_tmp/hwc_trace 8192 | grep -E "FRAMES|TRACE wave 0 frame 4" > gpurun_out/r2w/he_wave_trace.txt
cat gpurun_out/r2w/pmc_he_wave.txt | head -20
timeout 300 python3 tests/tools/esacf_kernel_times.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r2w/esacf_kernel_times.txt
cat gpurun_out/r2w/esacf_kernel_times.txt | cut -c1-250
