#!/bin/bash
# round 6, GPU call AE: stft_mel with its log-mel values stored one tile late (no vmcnt(0) behind the stores at every tile) against
# the library shipped until now (tools/variants/libwtalign_shipped.so): log-mel tests, alternating bench runs, kernel traces alone
# and in the pipelined region.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; out=$R/gpurun_out/r6ae; mkdir -p $out
NEW=$R/whisper-timestamped_amd/libwtalign.so; OLD=$R/tools/variants/libwtalign_shipped.so
sha256sum $OLD $NEW > $out/sha256_of_what_ran.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_guard.py tests/test_gpu_streams.py tests/test_pin_transformers.py -m gpu -q -k "logmel or mel or guard or pipeline or stream or padding" > $out/pytest_logmel.log 2>&1; echo "rc=$?" >> $out/pytest_logmel.log; grep -v "Warning\|warn" $out/pytest_logmel.log | tail -3
run() {  # name, lib, args...
  name=$1; lib=$2; shift; shift
  WT_LIBWTALIGN=$lib timeout 400 python3 bench.py --no-cpu-baseline --e2e off --other-configs off --steps 20 --warmup 5 "$@" > $out/$name.json 2> $out/$name.err
  python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
    print(json.dumps({"run": "$name", "ms_per_step": d['ms_per_step'], "min": d['timing']['ms_per_step_min'], "single": d['single_batch_in_flight']['ms_per_step'], "schedule": d['config'].get('schedule'), "stages": {k: v['ms'] for k, v in d['stages'].items()}, "parity": d['parity_in_leg'].get('ok'), "dlogmel": d['parity_in_leg'].get('max_abs_dlogmel'), "same": d.get('pipelined_equals_single_stream')}))
except Exception as e:
    print(json.dumps({"run": "$name", "error": repr(e)}))
PY
}
for rep in 1 2 3 4; do
  run kfull_old_$rep $OLD
  run kfull_new_$rep $NEW
done | tee $out/summary.jsonl
for rep in 1 2; do for wl in kfull256 largev3_fp16 kreal; do
  run ${wl}_old_$rep $OLD --workload $wl
  run ${wl}_new_$rep $NEW --workload $wl
done; done | tee -a $out/summary.jsonl
cd /tmp && export TMPDIR=/tmp
for v in old new; do
  lib=$OLD; [ $v = new ] && lib=$NEW
  WT_LIBWTALIGN=$lib timeout 300 rocprofv3 --kernel-trace --stats -d $out/kt_$v -o kt -- python $R/tools/run_pipelined_once.py kfull 200 > $out/kt_$v.log 2>&1
  grep "ms per step" $out/kt_$v.log
  python $R/tools/rocpd_stats.py $(find $out/kt_$v -name "*.db" | head -1) --skip 2 > $out/kernel_stats_pipelined_$v.txt 2>&1
  echo "== pipelined $v"; grep "dtw_kernel\|logprob_gather\|stft_mel\|rowmean" $out/kernel_stats_pipelined_$v.txt | cut -c1-130
  WT_LIBWTALIGN=$lib timeout 300 rocprofv3 --kernel-trace --stats -d $out/ks_$v -o kt -- python $R/bench.py --workload kfull --role kernel --pipeline 1 --steps 10 --warmup 2 --repeats 5 > $out/ks_$v.log 2>&1
  python $R/tools/rocpd_stats.py $(find $out/ks_$v -name "*.db" | head -1) --skip 2 > $out/kernel_stats_single_$v.txt 2>&1
  echo "== single $v"; grep "stft_mel\|logmel_finalize" $out/kernel_stats_single_$v.txt | cut -c1-130
done
find $out -name "*.db" -delete; find $out -name "*.csv" -size +1M -delete
