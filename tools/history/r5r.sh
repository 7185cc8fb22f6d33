#!/bin/bash
# round 5, GPU call R: cost and log-prob gather on streams of their own (they do not depend on each other), fresh processes.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$ROOT/gpurun_out/r5r
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for spec in "hilo 2" "hilo_lp_own 2" "prio3 2" "hilo_lp_own 1" "prio3 1" "prio3 3"; do
    set -- $spec
    timeout 200 python $ROOT/bench.py --role kernel --schedule $1 --pipeline $2 --out "$out/k_$1_p$2_r$rep.json" > "$out/k.log" 2>&1
    python -c "
import json
try:
    d=json.load(open('$out/k_$1_p$2_r$rep.json')); print(json.dumps({'schedule':'$1','batches_in_flight':$2,'rep':$rep,'ms_per_step':d['ms_per_step'],'min':d['timing']['ms_per_step_min'],'single':d['single_batch_in_flight']['ms_per_step'],'parity_ok':d['parity_in_leg']['ok']}))
except Exception as e: print('failed $1 $2', e)"
  done
done | tee "$out/schedules_cost_and_logprob_apart.jsonl"
