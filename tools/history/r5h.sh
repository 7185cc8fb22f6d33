#!/bin/bash
# round 5, GPU call H: the default-strategy leg with the worker-process sub-leg.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$ROOT/gpurun_out/r5h
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
t0=$(date +%s)
timeout 900 python $ROOT/bench.py --role e2e --leg efficient --no-cpu-baseline --out "$out/efficient_leg.json" > "$out/efficient_leg.log" 2>&1; echo "efficient rc=$?"
echo "seconds: $(( $(date +%s) - t0 ))"
grep -v "inconsistent length\|outside of audio" "$out/efficient_leg.log" | tail -8
python -c "
import json; d=json.load(open('$out/efficient_leg.json'))['long_form_1h_islands']
for k in ('ragged','ragged_worker_processes'): print(k, json.dumps(d.get(k))[:700])"
