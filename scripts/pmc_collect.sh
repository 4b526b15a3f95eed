#!/bin/bash
# usage (on the GPU box, from the repo root): bash scripts/pmc_collect.sh <tag> [workloads]
# One rocprofv3 pass per counter group over scripts/pmc_workloads.py (counters never share a pass with tracing other than
# --kernel-trace), then scripts/pmc_report.py -> gpurun_out/<tag>/pmc.json + pmc.txt; a --kernel-trace --stats pass first.
TAG=${1:-pmc}; WL=${2:-he,he_default,esacf_clips,esacf_1023,esacf_stft,prime,if0_clips,if0_stream}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pw_stats; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pw_stats -- python3 $GRAFT_REPO_ROOT/scripts/pmc_workloads.py $WL > $OUT/stats.log 2>&1
cp $(find /tmp/pw_stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
find /tmp/pw_stats -type f | head -20 >> $OUT/stats.log
cp $(find /tmp/pw_stats -name "*kernel_trace.csv" | head -1) $OUT/kernel_trace.csv
grep "pmc_workloads order:" $OUT/stats.log | sed "s/.*order: //" > $OUT/order.txt
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
         "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA"; do
  i=$((i+1)); rm -rf /tmp/pw_$i
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pw_$i -- python3 $GRAFT_REPO_ROOT/scripts/pmc_workloads.py $WL > $OUT/pass$i.log 2>&1
  f=$(find /tmp/pw_$i -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "pass $i ($C): no csv"; tail -3 $OUT/pass$i.log; else cp $f $OUT/pass$i.csv; fi
done
cd $GRAFT_REPO_ROOT
python3 scripts/pmc_report.py $OUT
rm -f $OUT/pass*.csv $OUT/kernel_trace.csv
