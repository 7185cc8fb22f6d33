#!/bin/bash
# round 5, GPU call G: hilo with ONE shared low-priority stream for the HBM-bound kernels of both batches, fresh processes.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$ROOT/gpurun_out/r5g
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for spec in "hilo 2" "hilo_one_lo 2" "hilo_one_lo_normal 2" "hilo_one_lo 3" "hilo_one_lo 4"; do
    set -- $spec
    timeout 200 python $ROOT/bench.py --role kernel --schedule $1 --pipeline $2 --out "$out/k_$1_p$2_r$rep.json" > "$out/k.log" 2>&1
    python - "$out/k_$1_p$2_r$rep.json" $1 $2 $rep <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(json.dumps({"schedule":sys.argv[2],"batches_in_flight":sys.argv[3],"rep":sys.argv[4],"ms_per_step":d["ms_per_step"],"min":d["timing"]["ms_per_step_min"],"max":d["timing"]["ms_per_step_max"],"single":d["single_batch_in_flight"]["ms_per_step"],"parity_ok":d["parity_in_leg"]["ok"]}))
except Exception as e: print("failed", sys.argv[1:], e); print(open(sys.argv[1].rsplit('/',1)[0]+'/k.log').read()[-1500:])
PY
  done
done | tee "$out/schedules_shared_lo.jsonl"
