#!/bin/bash
# Weak-scaling sweep of bench.py on ONE node: N = 1, 2, 4, 8 (or the list given), one process per GPU over RCCL, exactly as
# the driver launches it.  One JSON object per N on stdout: n_gpus, value, ms_per_step, the per-repeat timings, the
# one-in-flight value, the steps-only value and the gather's own milliseconds (bench.py reads the clock on both sides of the
# job's one all_gather), the EFFICIENCY against the N = 1 launch of this sweep -- value_N / (N x value_1), for the headline,
# for the headline without its gather and for every workload -- and, for N = 1, a check that the launcher path gives the
# plain `python3 bench.py` value.
#   bash scripts/scale_1to8.sh [--steps K] [--warmup W] [--gpus "1 2 4 8"] [--workloads LIST|--headline-only]
# Nothing here touches the GPU before torch.distributed.run starts the ranks (no exec after HIP initialisation); the CPU
# baseline legs run only at N = 1, inside bench.py, before its first HIP call.
# No multi-GPU node was available to the builder: this script has run at N = 1 only (and at N = 2 over gloo on CPU through
# tests/test_multirank_cpu.py, which launches bench.py the same way).
set -u
STEPS=2000; WARMUP=200; GPUS="1 2 4 8"; EXTRA="--headline-only"; PORT=${MASTER_PORT:-29517}
while [ $# -gt 0 ]; do
  case "$1" in
    --steps) STEPS=$2; shift 2;;
    --warmup) WARMUP=$2; shift 2;;
    --gpus) GPUS=$2; shift 2;;
    --workloads) EXTRA="--workloads $2"; shift 2;;
    --headline-only) EXTRA="--headline-only"; shift;;
    --all-workloads) EXTRA=""; shift;;
    *) echo "unknown argument $1" >&2; exit 2;;
  esac
done
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
# (SCALE_ASSUME_GPUS: tests only -- with bench.py's stand-in engine, MPX_BENCH_STUB, the ranks run over gloo on CPU)
HAVE=${SCALE_ASSUME_GPUS:-$(python3 -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 0)}
PLAIN=""
BASE=$(mktemp)   # the N = 1 launcher record of this sweep: what every later N is divided by
for N in $GPUS; do
  if [ "$N" -gt "$HAVE" ]; then echo "{\"n_gpus\": $N, \"skipped\": \"only $HAVE GPU(s) visible\"}"; continue; fi
  LOG=$(mktemp)
  if [ "$N" -eq 1 ]; then
    python3 bench.py --gpus 1 --steps $STEPS --warmup $WARMUP --no-cpu-baseline $EXTRA > $LOG 2> $LOG.err
    PLAIN=$(python3 -c "import json,sys; print(json.loads(open('$LOG').read().strip().splitlines()[-1])['value'])" 2>/dev/null || echo "")
  fi
  python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
      bench.py --gpus $N --steps $STEPS --warmup $WARMUP --no-cpu-baseline $EXTRA > $LOG 2> $LOG.err
  RC=$?
  python3 - "$LOG" "$N" "$RC" "$PLAIN" "$BASE" <<'PY'
import json, os, sys
log, n, rc, plain, base_path = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
lines = [l for l in open(log).read().splitlines() if l.startswith("{")]
if rc != 0 or not lines:
    print(json.dumps({"n_gpus": n, "failed": rc, "stderr_tail": open(log + ".err").read()[-800:]}))
    sys.exit(0)
d = json.loads(lines[-1])
out = {"n_gpus": d["n_gpus"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
       "ms_per_step_repeats": d.get("ms_per_step_repeats"), "value_one_in_flight": d.get("value_one_in_flight"),
       "per_gpu": d["value"] / d["n_gpus"], "scaling": d["scaling"], "steps": d["steps"],
       "value_steps_only": d.get("value_steps_only"), "gather_ms": d.get("gather_ms"),
       "gather": "one all_gather of the [steps, 12] sums inside the timed region (bench.py); value_steps_only stops the clock before it",
       "workloads": {k: {kk: v.get(kk) for kk in ("value", "unit", "ms", "scaling")} for k, v in d.get("workloads", {}).items()}}
if n == 1:
    with open(base_path, "w") as fh:
        json.dump(out, fh)
base = json.load(open(base_path)) if os.path.getsize(base_path) else None
if base:
    # `value` is whole-job throughput for weak (work grows with N) and strong (work fixed) workloads alike, so the ideal is
    # N x the one-GPU value in both cases
    eff = lambda vn, v1: (vn / (n * v1)) if vn and v1 else None
    out["efficiency_vs_n1"] = eff(out["value"], base["value"])
    out["efficiency_vs_n1_steps_only"] = eff(out.get("value_steps_only"), base.get("value_steps_only"))
    for k, w in out["workloads"].items():
        w["efficiency_vs_n1"] = eff(w.get("value"), (base["workloads"].get(k) or {}).get("value"))
if n == 1 and plain:
    out["plain_bench_value"] = float(plain)
    out["launcher_equals_plain_within_5pct"] = abs(float(plain) - d["value"]) <= 0.05 * d["value"]
print(json.dumps(out))
PY
  PORT=$((PORT+1))
done
