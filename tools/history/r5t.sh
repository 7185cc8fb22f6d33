#!/bin/bash
# round 5, GPU call T: the last state -- the streams / guard GPU tests touched since r5q, then the driver's exact command.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$ROOT/gpurun_out/r5t
mkdir -p "$out"
cd $ROOT
sha256sum whisper-timestamped_amd/libwtalign.so bench.py > "$out/head.txt"
timeout 900 python -m pytest tests/test_gpu_streams.py tests/test_gpu_streams_batch.py tests/test_gpu_guard.py -m gpu -q > "$out/pytest_gpu_streams_guard.log" 2>&1; echo "pytest rc=$?"; tail -3 "$out/pytest_gpu_streams_guard.log"
t0=$(date +%s)
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?"
echo "driver command wall seconds: $(( $(date +%s) - t0 ))" | tee "$out/bench.time"
python -c "
import json
d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['schedule'], d['roofline']['frac'], d['roofline']['traffic_source'], d.get('e2e_parity_failures'), d['faulted'])"
