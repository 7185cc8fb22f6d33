#!/bin/bash
# round 6, GPU call X: the row pass with 16-byte row stores (six 1 KB stores per 1500-frame row instead of 24 of 256 B) against the
# library shipped until now (tools/variants/libwtalign_shipped.so = d3b1ed67...): bit identity of wt_cost_batch, the parity tests,
# kernel traces of both, alternating bench runs.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; out=$R/gpurun_out/r6x; mkdir -p $out
NEW=$R/whisper-timestamped_amd/libwtalign.so; OLD=$R/tools/variants/libwtalign_shipped.so
sha256sum $OLD $NEW > $out/sha256_of_what_ran.txt
WT_LIBWTALIGN=$OLD timeout 300 python tools/ab_cost_bits.py > $out/bits_old.json 2> $out/bits_old.err
WT_LIBWTALIGN=$NEW timeout 300 python tools/ab_cost_bits.py > $out/bits_new.json 2> $out/bits_new.err
python - <<PY
import json
a=json.load(open('$out/bits_old.json')); b=json.load(open('$out/bits_new.json'))
same=[k for k in a if k!='lib' and a[k]==b.get(k)]; diff=[k for k in a if k!='lib' and a[k]!=b.get(k)]
print(json.dumps({"bit_identical_batches": len(same), "different": diff, "libs": [a['lib'], b['lib']]}))
PY
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_guard.py tests/test_gpu_streams.py -m gpu -q > $out/pytest_kernels.log 2>&1; echo "rc=$?" >> $out/pytest_kernels.log; grep -v "Warning\|warn" $out/pytest_kernels.log | tail -3
run() {  # name, lib, args...
  name=$1; lib=$2; shift; shift
  WT_LIBWTALIGN=$lib timeout 400 python3 bench.py --no-cpu-baseline --e2e off --other-configs off --steps 20 --warmup 5 "$@" > $out/$name.json 2> $out/$name.err
  python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
    print(json.dumps({"run": "$name", "ms_per_step": d['ms_per_step'], "single": d['single_batch_in_flight']['ms_per_step'], "schedule": d['config'].get('schedule'), "stages": {k: v['ms'] for k, v in d['stages'].items()}, "roofline": d['roofline']['frac'], "parity": d['parity_in_leg'].get('ok')}))
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
  for wl in kfull largev3_fp16; do
  WT_LIBWTALIGN=$lib timeout 300 rocprofv3 --kernel-trace --stats -d $out/kt_${wl}_$v -o kt -- python $R/bench.py --workload $wl --role kernel --pipeline 1 --steps 10 --warmup 2 --repeats 5 > $out/kt_${wl}_$v.log 2>&1
  python $R/tools/rocpd_stats.py $(find $out/kt_${wl}_$v -name "*.db" | head -1) --skip 2 > $out/kernel_stats_${wl}_$v.txt 2>&1
  echo "== $wl $v"; grep "rowmean\|colnorm" $out/kernel_stats_${wl}_$v.txt | cut -c1-140
  done
done
find $out -name "*.db" -delete; find $out -name "*.csv" -size +1M -delete
