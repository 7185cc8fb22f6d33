#!/bin/bash
# round 5, GPU call E: the driver's exact command on the new defaults (hilo schedule, ragged legs), then the kernel
# timeline of the hilo schedule.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$ROOT/gpurun_out/r5e
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
/usr/bin/time -v -o "$out/bench.time" timeout 1500 python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?"
grep "Elapsed" "$out/bench.time"
timeout 300 rocprofv3 --kernel-trace -d "$out/kt_hilo" -o kt -- python $ROOT/bench.py --role kernel --schedule hilo --pipeline 2 --steps 20 --warmup 3 --repeats 5 > "$out/kt_hilo.log" 2>&1
python $ROOT/tools/overlap_timeline.py "$(find "$out/kt_hilo" -name "*.db" | head -1)" --tail 0.4 > "$out/timeline_hilo.txt" 2>&1
find "$out" -name "*.db" -delete; find "$out" -name "*.csv" -delete
cat "$out/timeline_hilo.txt"; tail -c 1500 "$out/bench.err"
