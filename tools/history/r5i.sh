#!/bin/bash
# round 5, GPU call I: the new GPU tests (fenced digest entry, stream-priority schedules).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$ROOT/gpurun_out/r5i
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout 900 python -m pytest $ROOT/tests/test_gpu_guard.py $ROOT/tests/test_gpu_streams.py -m gpu -q -k "digest or schedules" > "$out/pytest_new.log" 2>&1; echo "pytest rc=$?"; tail -15 "$out/pytest_new.log"
