#!/bin/bash
# round 5, GPU call C: the -m gpu suite again (StreamRings without a tokenizer fixed), where the ragged B-stream word
# times differ from the one-stream ones, the default-strategy leg.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$ROOT/gpurun_out/r5c
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout 1200 python -m pytest $ROOT/tests -m gpu -q > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -5 "$out/pytest_gpu.log"
timeout 600 python $ROOT/tools/diag_ragged_parity.py 32 > "$out/diag_ragged_parity.txt" 2>&1; echo "diag rc=$?"
grep -v "inconsistent length\|outside of audio" "$out/diag_ragged_parity.txt" | tail -15
