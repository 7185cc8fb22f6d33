#!/bin/bash
# round 5, GPU call K: hilo vs hilo_cost_first, six interleaved fresh-process runs each.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$ROOT/gpurun_out/r5k
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3 4 5 6; do
  for s in hilo hilo_cost_first; do
    timeout 200 python $ROOT/bench.py --role kernel --schedule $s --out "$out/k_${s}_r$rep.json" > "$out/k.log" 2>&1
    python -c "
import json; d=json.load(open('$out/k_${s}_r$rep.json')); print(json.dumps({'schedule':'$s','rep':$rep,'ms_per_step':d['ms_per_step'],'min':d['timing']['ms_per_step_min'],'single':d['single_batch_in_flight']['ms_per_step']}))"
  done
done | tee "$out/hilo_vs_cost_first.jsonl"
