#!/bin/bash
# round 5, GPU call J: hilo issue orders, and stft_mel with 3 / 2 resident workgroups per CU (LDS room for the row pass beside it).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$ROOT/gpurun_out/r5j
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for spec in "hilo base" "hilo_cost_first base" "hilo_logmel_last base" "hilo lm_res3" "hilo lm_res2" "hilo_cost_first lm_res2" "serial lm_res2"; do
    set -- $spec
    if [ $2 = base ]; then unset WT_LIBWTALIGN; else export WT_LIBWTALIGN=$ROOT/tools/variants/libwtalign_$2.so; fi
    timeout 200 python $ROOT/bench.py --role kernel --schedule $1 --out "$out/k_$1_$2_r$rep.json" > "$out/k.log" 2>&1
    python - "$out/k_$1_$2_r$rep.json" $1 $2 $rep <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(json.dumps({"schedule":sys.argv[2],"library":sys.argv[3],"rep":sys.argv[4],"ms_per_step":d["ms_per_step"],"min":d["timing"]["ms_per_step_min"],"single":d["single_batch_in_flight"]["ms_per_step"],"logmel_stage_ms":d["stages"]["logmel"]["ms"],"parity_ok":d["parity_in_leg"]["ok"]}))
except Exception as e: print("failed", sys.argv[1:], e)
PY
  done
done | tee "$out/hilo_orders_and_stft_residency.jsonl"
