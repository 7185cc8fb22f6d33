#!/bin/bash
# round 5, GPU call B: the whole -m gpu suite on the new data plane (logits digests), the extended schedule matrix, the
# cost of a per-tile release fence + ticket in stft_mel (what a last-arriver finalisation would pay), the B-stream host
# profile, the default-strategy leg with its ragged legs.   -> gpurun_out/r5b/
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$ROOT/gpurun_out/r5b
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout 900 python -m pytest $ROOT/tests -m gpu -x -q > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -3 "$out/pytest_gpu.log"
timeout 600 python $ROOT/tools/overlap_matrix.py --out "$out/overlap_matrix.json" > "$out/overlap_matrix.log" 2>&1; echo "matrix rc=$?"
for v in base lm_ticket lm_fence; do
  if [ $v = base ]; then lib=""; else lib=$ROOT/tools/variants/libwtalign_$v.so; fi
  WT_LIBWTALIGN=$lib WT_AB_LABEL=$v timeout 300 python $ROOT/tools/ab_logmel.py >> "$out/ab_logmel_fence.jsonl" 2>> "$out/ab_logmel_fence.err"
done
timeout 600 python $ROOT/tools/profile_streams.py 256 > "$out/profile_streams_256.txt" 2>&1; echo "profile256 rc=$?"
timeout 600 python $ROOT/tools/profile_streams.py 128 > "$out/profile_streams_128.txt" 2>&1; echo "profile128 rc=$?"
timeout 900 python $ROOT/bench.py --role e2e --leg efficient --out "$out/efficient_leg.json" > "$out/efficient_leg.log" 2>&1; echo "efficient rc=$?"
grep -h "^B=" "$out"/profile_streams_*.txt; tail -4 "$out/overlap_matrix.log" | head -3; cat "$out/ab_logmel_fence.jsonl" | head -30
