#!/bin/bash
# GPU session C of round 3: full GPU test suite after the core refactor + fused small-unit kernel; kreal split vs fused.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; out=$R/gpurun_out/r3c; mkdir -p $out
python -c "import torch; torch.zeros(1, device='cuda')"
timeout 1500 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "rc=$?" >> $out/pytest_gpu.log
tail -15 $out/pytest_gpu.log
for a in split fused split fused; do
  timeout 300 python bench.py --workload kreal --align $a --e2e off --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'align':'$a','ms_per_step':d['ms_per_step'],'single':d['single_batch_in_flight']['ms_per_step'],'stages_ms':{k:v['ms'] for k,v in d['stages'].items()}}))"
done > $out/kreal_split_vs_fused.jsonl 2>&1
cat $out/kreal_split_vs_fused.jsonl
timeout 300 python bench.py --e2e off --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null > $out/bench_kfull.json; head -c 400 $out/bench_kfull.json
