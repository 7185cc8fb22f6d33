#!/bin/bash
# GPU session D of round 3: hybrid fused tail kernel -- parity tests, fenced buffers, kreal split vs fused.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; out=$R/gpurun_out/r3d; mkdir -p $out
python tools/dbg_fused.py 2>&1 | grep -v amdgpu.ids | head -20
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_guard.py -q -x > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log
tail -6 $out/pytest.log
for a in split fused split fused; do
  timeout 300 python bench.py --workload kreal --align $a --e2e off --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'align':'$a','ms_per_step':d['ms_per_step'],'single':d['single_batch_in_flight']['ms_per_step'],'stages_ms':{k:v['ms'] for k,v in d['stages'].items()}}))"
done > $out/kreal_split_vs_fused.jsonl 2>&1
cat $out/kreal_split_vs_fused.jsonl
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof_kreal -o kreal -- python $R/bench.py --workload kreal --align fused --e2e off --no-cpu-baseline --pipeline 1 --steps 20 --warmup 3 --repeats 2 > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(find $out/prof_kreal -name "*.db" | head -1) 2>/dev/null | head -20 > $out/kreal_fused_kernel_stats.txt || true
find $out/prof_kreal -name "*kernel_stats.csv" | head -1 | xargs -r head -12
cat $out/kreal_fused_kernel_stats.txt | head -14
find $out -name "*.db" -delete
