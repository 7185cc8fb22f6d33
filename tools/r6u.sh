#!/bin/bash
# round 6, GPU call U: the log-prob gather with the next four loads in flight while the current sixteen logits are reduced,
# taken only by launches of >= 16384 rows (tools/variants/libwtalign_lppipe.so), against the shipped library at 256 units
# (serial schedule, two buffer sets in flight) on ONE box, alternating processes; kfull as the control (its launch is below the switch).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; out=$R/gpurun_out/r6u; mkdir -p $out
NEW=$R/tools/variants/libwtalign_lppipe.so; OLD=$R/whisper-timestamped_amd/libwtalign.so
sha256sum $OLD $NEW > $out/sha256_of_what_ran.txt
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
for rep in 1 2 3; do for wl in kfull256 largev3_fp16; do
  run ${wl}_old_$rep $OLD --workload $wl
  run ${wl}_new_$rep $NEW --workload $wl
done; done | tee $out/summary.jsonl
for rep in 1 2; do for wl in kfull256; do
  run ${wl}_hilo_old_$rep $OLD --workload $wl --schedule hilo --sub-batches 4
  run ${wl}_hilo_new_$rep $NEW --workload $wl --schedule hilo --sub-batches 4
done; done | tee -a $out/summary.jsonl
run kfull_old_1 $OLD | tee -a $out/summary.jsonl
run kfull_new_1 $NEW | tee -a $out/summary.jsonl
WT_LIBWTALIGN=$NEW timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "logprob" > $out/pytest_logprob_variant.log 2>&1; echo "rc=$?" >> $out/pytest_logprob_variant.log; grep -v "Warning\|warn" $out/pytest_logprob_variant.log | tail -3
