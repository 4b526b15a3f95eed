#!/bin/bash
# usage (on the GPU box, from the repo root): bash scripts/evidence_refresh.sh <tag>
# Everything profiles/rN/ holds, collected in one go into gpurun_out/<tag>/: the GPU suite, the PMC passes
# (scripts/pmc_collect.sh) and their traffic table, bench.py plain (compact line + full record) / under rocprofv3 --kernel-trace --stats / one stream /
# at the driver's short step count, the stage-by-stage ESACF bit check and the host-copy probe (the last two need the
# dev library: `make -C chord-detection_amd/csrc dev` before the snapshot is taken).
TAG=${1:-refresh}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
DEV=$R/chord-detection_amd/libmpx_hip_dev.so
cd $R
timeout 1500 python3 -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/gputest.txt; cat $O/gputest.txt
export PMC_REPS=1
bash scripts/pmc_collect.sh $TAG/pmc_after > $O/pmc.log 2>&1
python3 scripts/pmc_to_traffic.py $O/pmc_after/pmc.json ${ROUND:-6} > /dev/null
cp profiles/traffic_latest.json $O/traffic_latest.json
timeout 900 python3 bench.py --steps 2000 --warmup 200 --full-json $O/bench_plain_full.json > $O/bench_plain.json 2> $O/bench_plain.err
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/kp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -- python3 $R/bench.py --steps 2000 --warmup 200 --no-cpu-baseline --full-json /tmp/bench_profiled_full.json > $O/bench_profiled.json 2> /tmp/kp.err
cp $(find /tmp/kp -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
rm -rf /tmp/kp1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp1 -- python3 $R/bench.py --steps 2000 --warmup 200 --streams 1 --no-cpu-baseline --headline-only --full-json /tmp/bench_s1_full.json > $O/bench_streams1_profiled.json 2> /tmp/kp1.err
cp $(find /tmp/kp1 -name "*kernel_stats.csv" | head -1) $O/bench_streams1_kernel_stats.csv
cd $R
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --full-json $O/bench_driver_short_full.json > $O/bench_driver_short.json 2> /dev/null
if [ -f $DEV ]; then
  MPX_LIB_PATH=$DEV timeout 900 python3 tests/tools/esacf_bitcheck.py 2>&1 | grep -v amdgpu > $O/esacf_bitcheck.txt; tail -1 $O/esacf_bitcheck.txt
  MPX_LIB_PATH=$DEV timeout 300 python3 scripts/h2d_probe.py > $O/h2d_probe.json 2> /dev/null
fi
if [ -f $R/chord-detection_amd/libmpx_hip_per3.so ]; then bash scripts/dev/spill_ab.sh run > $O/spill_ab.txt 2>&1; fi
head -c 300 $O/bench_driver_short.json
