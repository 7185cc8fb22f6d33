#!/bin/bash
# round 5, GPU call A: the new transformers pins on the GPU, the schedule matrix of the K-full step, the timeline of the
# two-batches-in-flight run, where a 256-stream decoder loop spends its host time.   -> gpurun_out/r5a/
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$ROOT/gpurun_out/r5a
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout 600 python -m pytest $ROOT/tests/test_pin_transformers.py -m gpu -x -q > "$out/pytest_pin.log" 2>&1; echo "pin rc=$?"
timeout 600 python $ROOT/tools/overlap_matrix.py --out "$out/overlap_matrix.json" > "$out/overlap_matrix.log" 2>&1; echo "matrix rc=$?"
timeout 300 rocprofv3 --kernel-trace -d "$out/kt_pipe" -o kt -- python $ROOT/bench.py --role kernel --pipeline 2 --steps 20 --warmup 3 --repeats 5 > "$out/kt_pipe.log" 2>&1
python $ROOT/tools/overlap_timeline.py "$(find "$out/kt_pipe" -name "*.db" | head -1)" --tail 0.4 > "$out/timeline_pipeline2.txt" 2>&1
timeout 300 rocprofv3 --kernel-trace -d "$out/kt_pipe3" -o kt -- python $ROOT/bench.py --role kernel --pipeline 3 --steps 21 --warmup 3 --repeats 5 > "$out/kt_pipe3.log" 2>&1
python $ROOT/tools/overlap_timeline.py "$(find "$out/kt_pipe3" -name "*.db" | head -1)" --tail 0.4 > "$out/timeline_pipeline3.txt" 2>&1
find "$out" -name "*.db" -delete; find "$out" -name "*.csv" -delete
timeout 600 python $ROOT/tools/profile_streams.py 256 > "$out/profile_streams_256.txt" 2>&1; echo "profile rc=$?"
tail -5 "$out/pytest_pin.log"; tail -12 "$out/overlap_matrix.log"; cat "$out/timeline_pipeline2.txt"
