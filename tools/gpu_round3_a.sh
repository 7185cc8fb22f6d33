#!/bin/bash
# GPU session A of round 3: import timing, the new fenced-buffer and multi-stream tests, A/B of the DTW plane-store forms,
# one run of the driver's bench command.  Writes gpurun_out/r3a/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; out=$R/gpurun_out/r3a; mkdir -p $out
t0=$(date +%s%3N); python -c "import torch" ; t1=$(date +%s%3N); python -c "import torch; torch.zeros(1, device='cuda')"; t2=$(date +%s%3N)
echo "{\"first_import_torch_ms\": $((t1-t0)), \"second_import_plus_cuda_init_ms\": $((t2-t1))}" > $out/import_time.json
timeout 900 python -m pytest tests/test_gpu_guard.py tests/test_gpu_streams.py -x -q > $out/pytest_guard_streams.log 2>&1; echo "rc=$?" >> $out/pytest_guard_streams.log
for l in asmstore new cstore asmstore new cstore; do
  if [ $l = new ]; then unset WT_LIBWTALIGN; else export WT_LIBWTALIGN=$R/tools/variants/libwtalign_$l.so; fi
  timeout 200 python bench.py --e2e off --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'lib':'$l','ms_per_step':d['ms_per_step'],'single':d['single_batch_in_flight']['ms_per_step'],'stages_ms':{k:v['ms'] for k,v in d['stages'].items()}}))"
done > $out/ab_plane_store.jsonl 2>&1
unset WT_LIBWTALIGN
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; echo "rc=$?" >> $out/bench_default.err
tail -3 $out/pytest_guard_streams.log; cat $out/ab_plane_store.jsonl; cat $out/import_time.json; head -c 600 $out/bench_default.json
