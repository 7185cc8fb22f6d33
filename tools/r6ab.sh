#!/bin/bash
# round 6, GPU call AB: ONLY fix00 (the 4 us launch that writes cost[0,0]) on the DTW stream instead of behind colnorm on the
# low-priority one (experiment library tools/variants/libwtalign_split.so with WT_SPLIT_COST=1: wt_cost_batch stops after the row
# pass, wt_dtw_batch runs the column pass first; same kernels, same bits) against the shipped library, alternating, one box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; out=$R/gpurun_out/r6ab; mkdir -p $out
NEW=$R/tools/variants/libwtalign_split.so; OLD=$R/whisper-timestamped_amd/libwtalign.so
sha256sum $OLD $NEW > $out/sha256_of_what_ran.txt
run() {  # name, lib, split, args...
  name=$1; lib=$2; split=$3; shift; shift; shift
  if [ "$split" = 1 ]; then export WT_SPLIT_COST=1; else unset WT_SPLIT_COST; fi
  WT_LIBWTALIGN=$lib timeout 400 python3 bench.py --no-cpu-baseline --e2e off --other-configs off --steps 20 --warmup 5 "$@" > $out/$name.json 2> $out/$name.err
  python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
    print(json.dumps({"run": "$name", "ms_per_step": d['ms_per_step'], "min": d['timing']['ms_per_step_min'], "single": d['single_batch_in_flight']['ms_per_step'], "schedule": d['config'].get('schedule'), "stages": {k: v['ms'] for k, v in d['stages'].items()}, "parity": d['parity_in_leg'].get('ok'), "same": d.get('pipelined_equals_single_stream')}))
except Exception as e:
    print(json.dumps({"run": "$name", "error": repr(e)}))
PY
}
for rep in 1 2 3 4; do
  run kfull_old_$rep $OLD 0
  run kfull_split_$rep $NEW 1
done | tee $out/summary.jsonl
