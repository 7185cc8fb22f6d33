#!/bin/bash
# round 5, GPU call D: the kernel-level leg under each --schedule in FRESH processes (the schedule matrix in one process
# showed that which HIP streams share a hardware queue decides the outcome), with 4 and 8 hardware queues; the B-stream
# host profile with and without the garbage collector paused; the default-strategy leg.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$ROOT/gpurun_out/r5d
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
 for q in default 8; do
  for sched in serial hilo two_streams dtw_hi; do
    if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    timeout 200 python $ROOT/bench.py --role kernel --schedule $sched --out "$out/k_${sched}_q${q}_r${rep}.json" > "$out/k.log" 2>&1
    python - "$out/k_${sched}_q${q}_r${rep}.json" $sched $q $rep <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(json.dumps({"schedule":sys.argv[2],"hw_queues":sys.argv[3],"rep":sys.argv[4],"ms_per_step":d["ms_per_step"],"min":d["timing"]["ms_per_step_min"],"max":d["timing"]["ms_per_step_max"],"single":d["single_batch_in_flight"]["ms_per_step"]}))
except Exception as e: print("failed", sys.argv[1:], e)
PY
  done
 done
done | tee "$out/schedules_fresh_processes.jsonl"
unset GPU_MAX_HW_QUEUES
for p in 3; do
  timeout 200 python $ROOT/bench.py --role kernel --schedule hilo --pipeline $p --out "$out/k_hilo_p$p.json" > "$out/k.log" 2>&1; python -c "import json;d=json.load(open('$out/k_hilo_p$p.json'));print('hilo pipeline $p', d['ms_per_step'])"
  timeout 200 python $ROOT/bench.py --role kernel --schedule two_streams --pipeline $p --out "$out/k_two_p$p.json" > "$out/k.log" 2>&1; python -c "import json;d=json.load(open('$out/k_two_p$p.json'));print('two_streams pipeline $p', d['ms_per_step'])"
done | tee -a "$out/schedules_fresh_processes.jsonl"
WT_PAUSE_GC=0 timeout 600 python $ROOT/tools/profile_streams.py 256 > "$out/profile_streams_256_gc_on.txt" 2>&1
timeout 600 python $ROOT/tools/profile_streams.py 256 > "$out/profile_streams_256_gc_paused.txt" 2>&1
WT_PAUSE_GC=0 timeout 600 python $ROOT/tools/profile_streams.py 128 > "$out/profile_streams_128_gc_on.txt" 2>&1
timeout 600 python $ROOT/tools/profile_streams.py 128 > "$out/profile_streams_128_gc_paused.txt" 2>&1
grep -h "^B=" "$out"/profile_streams_*.txt
timeout 900 python $ROOT/bench.py --role e2e --leg efficient --out "$out/efficient_leg.json" > "$out/efficient_leg.log" 2>&1; echo "efficient rc=$?"
grep -v "inconsistent length\|outside of audio" "$out/efficient_leg.log" | tail -5
