#!/bin/bash
# GPU session F: multi-wave fused tail kernel -- parity, fenced buffers, kreal split vs fused, kernel trace.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; out=$R/gpurun_out/r3f; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_guard.py tests/test_gpu_transcribe.py -q -x > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log
tail -6 $out/pytest.log
for a in split fused split fused; do
  timeout 300 python bench.py --workload kreal --align $a --e2e off --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'align':'$a','ms_per_step':d['ms_per_step'],'single':d['single_batch_in_flight']['ms_per_step'],'stages_ms':{k:v['ms'] for k,v in d['stages'].items()}}))"
done > $out/kreal_split_vs_fused.jsonl 2>&1
cat $out/kreal_split_vs_fused.jsonl
cd /tmp && export TMPDIR=/tmp
for a in fused split; do
timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof_kreal_$a -o kreal -- python $R/bench.py --role kernel --workload kreal --align $a --pipeline 1 --steps 20 --warmup 3 --repeats 2 > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(find $out/prof_kreal_$a -name "*.db" | head -1) --skip 3 2>/dev/null | grep -v "at6native" | head -14 > $out/kreal_${a}_kernel_stats.txt
cat $out/kreal_${a}_kernel_stats.txt | cut -c1-150
done
find $out -name "*.db" -delete; find $out -name "*.csv" -delete
