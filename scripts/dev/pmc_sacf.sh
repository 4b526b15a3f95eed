#!/bin/bash
cd /tmp; export TMPDIR=/tmp
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES"; do
  rm -rf /tmp/pmc_out
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_out -- python3 $GRAFT_REPO_ROOT/tests/tools/esacf_kernel_times.py > /tmp/pmc_log.txt 2>&1
  f=$(find /tmp/pmc_out -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "no csv for $C"; tail -5 /tmp/pmc_log.txt; continue; fi
  python3 - "$f" <<'PY'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n=r['Kernel_Name']
    for key in ('sacf_pfa_kernel<2','sacf_pfa_kernel<1','bandsplit','peakfit','coopfit'):
        if key in n: acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for key,d in acc.items():
    for k,v in d.items():
        v=sorted(v); print("%-18s %-24s median %.4g (n=%d)"%(key, k, v[len(v)//2], len(v)))
PY
done
