#!/bin/bash
# round 6, GPU call G: after the last code change (timing events recycled in pipeline.StageSet): the pipeline / guard / batched
# tests and the driver's bench command once more.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; out=$R/gpurun_out/r6g; mkdir -p $out
sha256sum whisper-timestamped_amd/libwtalign.so bench.py benchlib/*.py whisper-timestamped_amd/whisper_timestamped/*.py > $out/sha256_of_what_ran.txt
timeout 1200 python -m pytest tests/test_gpu_streams.py tests/test_gpu_guard.py tests/test_gpu_transcribe.py -m gpu -q -k "pipeline or batches or guard or batched or naive or step" > $out/pytest_subset.log 2>&1; echo "rc=$?" >> $out/pytest_subset.log; grep -v "Warning\|warn" $out/pytest_subset.log | tail -3
t0=$(date +%s)
timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_run1.json 2> $out/bench_run1.err; echo "{\"run\": 1, \"rc\": $?, \"wall_s\": $(( $(date +%s) - t0 ))}" | tee $out/bench_runs.jsonl
python - <<PY
import json
d=json.loads(open('$out/bench_run1.json').read().strip().splitlines()[-1])
e=d.get('e2e',{})
print(d['value'], d['ms_per_step'], d['single_batch_in_flight']['ms_per_step'], d['roofline']['frac'], {k: v['ms'] for k, v in d['stages'].items()}, d.get('parity_failures'), [k for k in d if 'error' in k], [k for k in e if 'error' in k], e.get('audio_s_per_s'))
PY
