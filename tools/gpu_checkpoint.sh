#!/bin/bash
# Freeze-and-verify sequence (GPU box): the full GPU test suite, smoke(), then the driver's bench command N times from
# the repo directory, cold processes.  tools/gpu_checkpoint.sh <tag> [N]   -> gpurun_out/<tag>/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; tag=${1:-ckpt}; n=${2:-5}; out=$R/gpurun_out/$tag; mkdir -p $out
( cd $R && git rev-parse HEAD 2>/dev/null || true ) > $out/head.txt
sha256sum whisper-timestamped_amd/libwtalign.so bench.py >> $out/head.txt
timeout 1800 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "rc=$?" >> $out/pytest_gpu.log
tail -4 $out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "rc=$?" >> $out/smoke.log; tail -2 $out/smoke.log
for i in $(seq 1 $n); do
  timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_run$i.json 2> $out/bench_run$i.err; echo "{\"run\": $i, \"rc\": $?}" >> $out/bench_runs.jsonl
  python -c "
import json
d=json.loads(open('$out/bench_run$i.json').read().strip().splitlines()[-1])
print('run $i', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('cpu_baseline',{}).get('value'), d.get('e2e',{}).get('audio_s_per_s'), d.get('kernel_leg_attempts'), [k for k in d if 'error' in k])"
done
