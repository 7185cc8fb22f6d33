#!/bin/bash
# round 6, GPU call M: A/B on ONE box: the gather with the next four loads issued before the current sixteen logits are reduced (variant, -DWT_LP_PIPE) against the shipped one ("old")
# byte-identical to the library of commits up to ac0ea54) against the branch-free core; then the digest test.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; out=$R/gpurun_out/r6m; mkdir -p $out
sha256sum whisper-timestamped_amd/libwtalign.so tools/variants/libwtalign_pipe.so > $out/sha256_of_what_ran.txt
run() {  # name, lib, args...
  name=$1; lib=$2; shift; shift
  WT_LIBWTALIGN=$lib timeout 400 python3 bench.py --no-cpu-baseline --e2e off --other-configs off --steps 20 --warmup 5 "$@" > $out/$name.json 2> $out/$name.err
  python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
    print(json.dumps({"run": "$name", "ms_per_step": d['ms_per_step'], "single": d['single_batch_in_flight']['ms_per_step'], "schedule": d['config'].get('schedule'), "stages": {k: v['ms'] for k, v in d['stages'].items()}, "roofline": d['roofline']['frac'], "parity": d['parity_in_leg'].get('ok'), "dlogprob": d['parity_in_leg'].get('max_abs_dlogprob')}))
except Exception as e:
    print(json.dumps({"run": "$name", "error": repr(e)}))
PY
}
NEW=$R/tools/variants/libwtalign_pipe.so; OLD=$R/whisper-timestamped_amd/libwtalign.so
for rep in 1 2 3 4; do
  run kfull_old_$rep $OLD
  run kfull_new_$rep $NEW
done | tee $out/summary.jsonl
for rep in 1 2; do for wl in kfull256 largev3_fp16 kreal; do
  run ${wl}_old_$rep $OLD --workload $wl
  run ${wl}_new_$rep $NEW --workload $wl
done; done | tee -a $out/summary.jsonl
