#!/bin/bash
# GPU session B of round 3.  Writes gpurun_out/r3b/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; out=$R/gpurun_out/r3b; mkdir -p $out
python -c "import torch; torch.zeros(1, device='cuda')"
timeout 900 python -m pytest tests/test_gpu_guard.py tests/test_gpu_streams.py -q > $out/pytest_guard_streams.log 2>&1; echo "rc=$?" >> $out/pytest_guard_streams.log
tail -4 $out/pytest_guard_streams.log
# pageable D2H of GBs in cold processes, no libwtalign kernel at all
f=0; for i in $(seq 1 ${D2H_RUNS:-60}); do timeout 120 python tools/d2h_probe.py > $out/d2h.out 2> $out/d2h.err; rc=$?; if [ $rc -ne 0 ]; then f=$((f+1)); { echo "== run $i rc=$rc"; tail -3 $out/d2h.err; } >> $out/d2h_faults.txt; fi; done
echo "{\"cold_runs\": ${D2H_RUNS:-60}, \"non_zero_exits\": $f}" | tee $out/d2h_probe.json
# fp16 rowmean: previous build (registers) vs in-tree (LDS-DMA), largev3_fp16 workload
for l in prev new prev new; do
  if [ $l = new ]; then unset WT_LIBWTALIGN; else export WT_LIBWTALIGN=$R/tools/variants/libwtalign_prev.so; fi
  timeout 300 python bench.py --workload largev3_fp16 --e2e off --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'lib':'$l','ms_per_step':d['ms_per_step'],'single':d['single_batch_in_flight']['ms_per_step'],'stages':{k:(v['ms'],v['frac_hbm']) for k,v in d['stages'].items()}}))"
done > $out/ab_fp16_rowmean.jsonl 2>&1
unset WT_LIBWTALIGN
cat $out/ab_fp16_rowmean.jsonl
timeout 1200 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "rc=$?" >> $out/pytest_gpu.log
tail -5 $out/pytest_gpu.log
