#!/bin/bash
# round 6, GPU call A: the pipeline (whisper_timestamped/pipeline.py) as the bench's timed region and under batched.py:
# the new GPU tests, the kernel-level line, and the large-batch workloads launched as chunk ranges (sub-batches).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; out=$R/gpurun_out/r6a; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_streams.py tests/test_gpu_guard.py -m gpu -x -q > $out/pytest_streams.log 2>&1; echo "rc=$?" >> $out/pytest_streams.log; tail -5 $out/pytest_streams.log
timeout 900 python -m pytest tests/test_gpu_transcribe.py -m gpu -x -q -k "batched or naive" > $out/pytest_batched.log 2>&1; echo "rc=$?" >> $out/pytest_batched.log; tail -5 $out/pytest_batched.log
K="--e2e off --no-cpu-baseline --other-configs off --steps 20 --warmup 5"
for i in 1 2; do
  timeout 300 python bench.py $K > $out/kfull_run$i.json 2> $out/kfull_run$i.err
  python - <<PY
import json
d=json.loads(open('$out/kfull_run$i.json').read().strip().splitlines()[-1])
print('kfull run $i', d['value'], d['ms_per_step'], d['single_batch_in_flight']['ms_per_step'], d['roofline']['frac'], d['whole_step'], d['config']['schedule'], d['config']['alignment_entry'])
PY
done
for wl in kfull256 largev3_fp16; do
  for cfgs in "serial 1" "hilo 1" "hilo 2" "hilo 4" "hilo 8" "serial 4"; do
    set -- $cfgs
    timeout 300 python bench.py --workload $wl $K --min-seconds 0.5 --schedule $1 --sub-batches $2 > $out/${wl}_$1_$2.json 2> $out/${wl}_$1_$2.err
    python - <<PY
import json
try:
    d=json.loads(open('$out/${wl}_$1_$2.json').read().strip().splitlines()[-1])
    print('$wl $1 sub-batches $2:', d['ms_per_step'], 'single', d['single_batch_in_flight']['ms_per_step'], d['config']['schedule'], d['config']['sub_batches'], d['parity_in_leg']['ok'], {k: v['ms'] for k, v in d['stages'].items()})
except Exception as e:
    print('$wl $1 $2 failed', e)
PY
  done
done
