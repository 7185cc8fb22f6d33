#!/bin/bash
# round 6, GPU call N: the DTW sweep with a prefetch distance of TWO blocks (three block buffers): parity (DTW tests + stress), then
# A/B/C on one box, alternating: prev = the shipped library (d3b1ed67), dtw3 = + the deeper DTW prefetch, dtw3_pipe = + the gather
# that issues its next four loads before it reduces the current sixteen logits; then a Gantt of dtw3 and of dtw3_pipe.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; out=$R/gpurun_out/r6n; mkdir -p $out
sha256sum whisper-timestamped_amd/libwtalign.so tools/variants/*.so > $out/sha256_of_what_ran.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_guard.py -m gpu -q -x > $out/pytest_parity.log 2>&1; echo "rc=$?" >> $out/pytest_parity.log; grep -v "Warning\|warn" $out/pytest_parity.log | tail -3
ls tests/stress
timeout 600 python tests/stress/stress_dtw.py > $out/stress_dtw.log 2>&1; echo "rc=$?" >> $out/stress_dtw.log; tail -3 $out/stress_dtw.log
run() {  # name, lib, args...
  name=$1; lib=$2; shift; shift
  WT_LIBWTALIGN=$lib timeout 400 python3 bench.py --no-cpu-baseline --e2e off --other-configs off --steps 20 --warmup 5 "$@" > $out/$name.json 2> $out/$name.err
  python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
    print(json.dumps({"run": "$name", "ms_per_step": d['ms_per_step'], "single": d['single_batch_in_flight']['ms_per_step'], "schedule": d['config'].get('schedule'), "stages": {k: v['ms'] for k, v in d['stages'].items()}, "roofline": d['roofline']['frac'], "parity": d['parity_in_leg'].get('ok'), "dlogprob": d['parity_in_leg'].get('max_abs_dlogprob')}))
except Exception as e:
    print(json.dumps({"run": "$name", "error": repr(e)}))
PY
}
PREV=$R/tools/variants/libwtalign_prev.so; DTW3=$R/whisper-timestamped_amd/libwtalign.so; BOTH=$R/tools/variants/libwtalign_dtw3_pipe.so
for rep in 1 2 3; do
  run kfull_prev_$rep $PREV
  run kfull_dtw3_$rep $DTW3
  run kfull_dtw3pipe_$rep $BOTH
done | tee $out/summary.jsonl
for wl in kfull256 largev3_fp16 kreal; do
  run ${wl}_prev $PREV --workload $wl
  run ${wl}_dtw3 $DTW3 --workload $wl
  run ${wl}_dtw3pipe $BOTH --workload $wl
done | tee -a $out/summary.jsonl
cd /tmp && export TMPDIR=/tmp
for v in dtw3 dtw3pipe; do
  lib=$DTW3; [ $v = dtw3pipe ] && lib=$BOTH
  WT_LIBWTALIGN=$lib timeout 300 rocprofv3 --kernel-trace -d $out/tl_$v -o kt -- python $R/bench.py --role kernel --steps 20 --warmup 3 --repeats 6 --workload kfull > $out/tl_$v.log 2>&1
  python $R/tools/gantt.py $(find $out/tl_$v -name "*.db" | head -1) --steps 3 > $out/gantt_kfull_hilo_$v.txt 2>&1
  tail -4 $out/gantt_kfull_hilo_$v.txt
done
find $out -name "*.db" -delete; find $out -name "*.csv" -size +1M -delete
