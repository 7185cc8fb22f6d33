#!/bin/bash
# round 6, GPU call AF: stages take turns ACROSS the two buffer sets (WT_PIPE_TURNS): a log-prob gather also waits for the other
# set's previous gather, so that a gather shares the chip with the other set's cost stage instead of with its twin (the Gantt of
# the shipped schedule shows the sets in lock step: both gathers together, then both row passes).  Same kernels, same library.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; out=$R/gpurun_out/r6af; mkdir -p $out
run() {  # name, turns, args...
  name=$1; turns=$2; shift; shift
  if [ -n "$turns" ]; then export WT_PIPE_TURNS=$turns; else unset WT_PIPE_TURNS; fi
  timeout 400 python3 bench.py --no-cpu-baseline --e2e off --other-configs off --steps 20 --warmup 5 "$@" > $out/$name.json 2> $out/$name.err
  python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
    print(json.dumps({"run": "$name", "ms_per_step": d['ms_per_step'], "min": d['timing']['ms_per_step_min'], "single": d['single_batch_in_flight']['ms_per_step'], "schedule": d['config'].get('schedule'), "parity": d['parity_in_leg'].get('ok'), "same": d.get('pipelined_equals_single_stream')}))
except Exception as e:
    print(json.dumps({"run": "$name", "error": repr(e)}))
PY
}
for rep in 1 2 3; do
  run kfull_none_$rep ""
  run kfull_logprob_$rep logprob
  run kfull_cost_$rep cost
  run kfull_both_$rep logprob,cost
done | tee $out/summary.jsonl
for rep in 1 2; do for wl in kfull256 largev3_fp16; do
  run ${wl}_none_$rep "" --workload $wl
  run ${wl}_logprob_$rep logprob --workload $wl
  run ${wl}_both_$rep logprob,cost --workload $wl
done; done | tee -a $out/summary.jsonl
