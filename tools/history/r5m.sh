#!/bin/bash
# round 5, GPU call M: the 256-chunk workloads (kfull256, largev3_fp16 = BASELINE configs[4] shapes) and kreal under each schedule.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$ROOT/gpurun_out/r5m
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for wl in largev3_fp16 kfull256 kreal; do
  for s in serial hilo dtw_hi two_streams hilo_cost_first; do
    timeout 200 python $ROOT/bench.py --role kernel --workload $wl --schedule $s --min-seconds 0.5 --secondary --out "$out/k_${wl}_$s.json" > "$out/k.log" 2>&1
    python -c "
import json
try:
    d=json.load(open('$out/k_${wl}_$s.json')); print(json.dumps({'workload':'$wl','schedule':'$s','ms_per_step':d['ms_per_step'],'min':d['timing']['ms_per_step_min'],'single':d['single_batch_in_flight']['ms_per_step'],'parity_ok':d['parity_in_leg']['ok']}))
except Exception as e: print('failed $wl $s', e)"
  done
done | tee "$out/schedules_secondary_workloads.jsonl"
