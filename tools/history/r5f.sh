#!/bin/bash
# round 5, GPU call F: the driver's exact command on the new defaults.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$ROOT/gpurun_out/r5f
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
t0=$(date +%s)
timeout 1500 python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?"
echo "driver command wall seconds: $(( $(date +%s) - t0 ))" | tee "$out/bench.time"
grep -v "inconsistent length\|outside of audio" "$out/bench.err" | tail -c 1500
