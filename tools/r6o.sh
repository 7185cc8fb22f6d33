#!/bin/bash
# round 6, GPU call O: kfull, 2 batches in flight: schedules {hilo, hilo3 = the DTW on a high-priority stream of its own} x libraries
# {prev = shipped, dtw3 = DTW prefetch distance 2 blocks, pipe = gather issues its next four loads before reducing the current 16
# logits, dtw3_pipe = both}, alternating, twice.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; out=$R/gpurun_out/r6o; mkdir -p $out
sha256sum whisper-timestamped_amd/libwtalign.so tools/variants/*.so > $out/sha256_of_what_ran.txt
run() {  # name, lib, args...
  name=$1; lib=$2; shift; shift
  WT_LIBWTALIGN=$lib timeout 400 python3 bench.py --no-cpu-baseline --e2e off --other-configs off --steps 20 --warmup 5 "$@" > $out/$name.json 2> $out/$name.err
  python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
    print(json.dumps({"run": "$name", "ms_per_step": d['ms_per_step'], "single": d['single_batch_in_flight']['ms_per_step'], "schedule": d['config'].get('schedule'), "parity": d['parity_in_leg'].get('ok'), "same_as_single_stream": d.get('pipelined_equals_single_stream')}))
except Exception as e:
    print(json.dumps({"run": "$name", "error": repr(e)}))
PY
}
declare -A LIB=( [prev]=$R/tools/variants/libwtalign_prev.so [dtw3]=$R/whisper-timestamped_amd/libwtalign.so [pipe]=$R/tools/variants/libwtalign_pipe.so [dtw3pipe]=$R/tools/variants/libwtalign_dtw3_pipe.so )
for rep in 1 2; do for sch in hilo hilo3; do for v in prev dtw3 pipe dtw3pipe; do
  run kfull_${sch}_${v}_$rep ${LIB[$v]} --schedule $sch
done; done; done | tee $out/summary.jsonl
cd /tmp && export TMPDIR=/tmp
WT_LIBWTALIGN=${LIB[pipe]} timeout 300 rocprofv3 --kernel-trace -d $out/tl -o kt -- python $R/bench.py --role kernel --steps 20 --warmup 3 --repeats 6 --workload kfull --schedule hilo3 > $out/tl.log 2>&1
python $R/tools/gantt.py $(find $out/tl -name "*.db" | head -1) --steps 3 > $out/gantt_kfull_hilo3_pipe.txt 2>&1
cat $out/gantt_kfull_hilo3_pipe.txt
find $out -name "*.db" -delete; find $out -name "*.csv" -size +1M -delete
