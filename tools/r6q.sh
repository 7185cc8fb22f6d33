#!/bin/bash
# round 6, GPU call Q: the fp32 row pass with TWO head rows in flight per wave (WT_ROWMEAN_DEEP=1, tools/variants/libwtalign_deep.so)
# against the shipped library on ONE box: bit identity of wt_cost_batch (tools/ab_cost_bits.py), the cost tests on the
# variant, then alternating bench runs (kfull, kfull256).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; out=$R/gpurun_out/r6q; mkdir -p $out
NEW=$R/tools/variants/libwtalign_deep.so; OLD=$R/whisper-timestamped_amd/libwtalign.so
sha256sum $OLD $NEW > $out/sha256_of_what_ran.txt
WT_LIBWTALIGN=$OLD timeout 300 python tools/ab_cost_bits.py > $out/bits_old.json 2> $out/bits_old.err
WT_LIBWTALIGN=$NEW timeout 300 python tools/ab_cost_bits.py > $out/bits_new.json 2> $out/bits_new.err
python - <<PY
import json
a=json.load(open('$out/bits_old.json')); b=json.load(open('$out/bits_new.json'))
same=[k for k in a if k!='lib' and a[k]==b.get(k)]; diff=[k for k in a if k!='lib' and a[k]!=b.get(k)]
print(json.dumps({"bit_identical_batches": len(same), "different": diff, "libs": [a['lib'], b['lib']]}))
PY
WT_LIBWTALIGN=$NEW timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "cost or batch or alignment" > $out/pytest_cost_variant.log 2>&1; echo "rc=$?" >> $out/pytest_cost_variant.log; grep -v "Warning\|warn" $out/pytest_cost_variant.log | tail -3
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
for rep in 1 2 3; do
  run kfull_old_$rep $OLD
  run kfull_new_$rep $NEW
done | tee $out/summary.jsonl
for rep in 1 2; do
  run kfull256_old_$rep $OLD --workload kfull256
  run kfull256_new_$rep $NEW --workload kfull256
done | tee -a $out/summary.jsonl
cd /tmp && export TMPDIR=/tmp
for v in old new; do
  lib=$OLD; [ $v = new ] && lib=$NEW
  WT_LIBWTALIGN=$lib timeout 300 rocprofv3 --kernel-trace --stats -d $out/kt_$v -o kt -- python $R/bench.py --workload kfull --role kernel --pipeline 1 --steps 10 --warmup 2 --repeats 5 > $out/kt_$v.log 2>&1
  python $R/tools/rocpd_stats.py $(find $out/kt_$v -name "*.db" | head -1) --skip 2 > $out/kernel_stats_kfull_$v.txt 2>&1
  head -8 $out/kernel_stats_kfull_$v.txt
done
find $out -name "*.db" -delete; find $out -name "*.csv" -size +1M -delete
