#!/bin/bash
cd /tmp; export TMPDIR=/tmp
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES"; do
  rm -rf /tmp/pmc_out
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_out -- $GRAFT_REPO_ROOT/build_tmp/he_wave_check 8192 > /tmp/pmc_log.txt 2>&1
  f=$(find /tmp/pmc_out -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "no csv for $C"; tail -5 /tmp/pmc_log.txt; continue; fi
  python3 - "$f" <<'PY'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.defaultdict(list)
names=set()
for r in rows:
    names.add(r['Kernel_Name'][:60])
    if 'he_wave_kernel' in r['Kernel_Name'] and ('false' in r['Kernel_Name'] or 'Lb0' in r['Kernel_Name']):
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
if not acc: print(names)
for k,v in acc.items():
    v=sorted(v); print("%-24s median %.4g (n=%d)"%(k, v[len(v)//2], len(v)))
PY
done
