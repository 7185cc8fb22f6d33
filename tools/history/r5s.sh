#!/bin/bash
# round 5, GPU call S: wave priority (s_setprio 2) for the HBM-bound kernels (log-prob gather; + rowmean) under the hilo schedule.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$ROOT/gpurun_out/r5s
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do
  for v in base lp_prio2 lp_rm_prio2; do
    if [ $v = base ]; then unset WT_LIBWTALIGN; else export WT_LIBWTALIGN=$ROOT/tools/variants/libwtalign_$v.so; fi
    timeout 200 python $ROOT/bench.py --role kernel --out "$out/k_${v}_r$rep.json" > "$out/k.log" 2>&1
    python -c "
import json
try:
    d=json.load(open('$out/k_${v}_r$rep.json')); print(json.dumps({'library':'$v','rep':$rep,'ms_per_step':d['ms_per_step'],'min':d['timing']['ms_per_step_min'],'single':d['single_batch_in_flight']['ms_per_step'],'logprob_ms':d['stages']['logprob']['ms'],'cost_ms':d['stages']['cost']['ms'],'parity_ok':d['parity_in_leg']['ok']}))
except Exception as e: print('failed $v', e)"
  done
done | tee "$out/wave_priority_of_the_hbm_bound_kernels.jsonl"
