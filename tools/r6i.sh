#!/bin/bash
# round 6, GPU call I: the branch-free log-sum-exp core of logprob_gather / logprob_digest (3.7 VALU instructions per logit
# instead of 14.4): parity tests, then the kernel legs of every workload under both schedules.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; out=$R/gpurun_out/r6i; mkdir -p $out
sha256sum whisper-timestamped_amd/libwtalign.so > $out/sha256_of_what_ran.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_guard.py tests/test_gpu_streams_batch.py -m gpu -q -x -k "logprob or guard or digest or stream" > $out/pytest_logprob.log 2>&1; echo "rc=$?" >> $out/pytest_logprob.log; grep -v "Warning\|warn" $out/pytest_logprob.log | tail -4
run() {  # name, args...
  name=$1; shift
  timeout 400 python3 bench.py --no-cpu-baseline --e2e off --other-configs off --steps 20 --warmup 5 "$@" > $out/$name.json 2> $out/$name.err
  python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
    print(json.dumps({"run": "$name", "ms_per_step": d['ms_per_step'], "single": d['single_batch_in_flight']['ms_per_step'], "schedule": d['config'].get('schedule'), "stages": {k: v['ms'] for k, v in d['stages'].items()}, "roofline": d['roofline']['frac'], "parity": d['parity_in_leg'].get('ok'), "dlogprob": d['parity_in_leg'].get('max_abs_dlogprob')}))
except Exception as e:
    print(json.dumps({"run": "$name", "error": repr(e)}))
PY
}
for rep in 1 2; do
  run kfull_auto_$rep
  run kfull_serial_$rep --schedule serial
  for wl in kfull256 largev3_fp16; do
    run ${wl}_serial_$rep --workload $wl --schedule serial
    run ${wl}_hilo_$rep --workload $wl --schedule hilo
  done
done | tee $out/summary.jsonl
