#!/bin/bash
# round 6, GPU call K: the checkpoint after the branch-free log-sum-exp core (fc9a36a): full `pytest -m gpu`, smoke(), the driver's bench
# and the profile set of the kernel-level leg through the product pipeline (kernel trace, FETCH_SIZE / WRITE_SIZE passes),
# plus the timelines of the pipelined schedules: kfull under hilo, kfull256 serial vs hilo as 4 chunk ranges.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; out=$R/gpurun_out/r6k; mkdir -p $out
sha256sum whisper-timestamped_amd/libwtalign.so bench.py benchlib/*.py whisper-timestamped_amd/whisper_timestamped/*.py > $out/sha256_of_what_ran.txt
timeout 2400 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "rc=$?" >> $out/pytest_gpu.log; grep -v "Warning\|warn" $out/pytest_gpu.log | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "rc=$?" >> $out/smoke.log; tail -2 $out/smoke.log
for i in 1 2; do
  t0=$(date +%s)
  timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_run$i.json 2> $out/bench_run$i.err; echo "{\"run\": $i, \"rc\": $?, \"wall_s\": $(( $(date +%s) - t0 ))}" >> $out/bench_runs.jsonl
  python - <<PY
import json
d=json.loads(open('$out/bench_run$i.json').read().strip().splitlines()[-1])
print('run $i', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('whole_step'), d.get('cpu_baseline',{}).get('value'), d.get('e2e',{}).get('audio_s_per_s'), d.get('parity_failures'), d.get('max_abs_dt_word_vs_ref_s'), [k for k in d if 'error' in k], {k: (v.get('ms_per_step'), v.get('error')) for k, v in (d.get('other_configs') or {}).items()})
PY
done
tail -2 $out/bench_runs.jsonl
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
for wl in kfull kreal kfull256 largev3_fp16; do
  K="$B --workload $wl --role kernel --pipeline 1"
  timeout 300 rocprofv3 --kernel-trace --stats -d $out/kt_$wl -o kt -- $K --steps 10 --warmup 2 --repeats 5 > $out/kt_$wl.log 2>&1
  python $R/tools/rocpd_stats.py $(find $out/kt_$wl -name "*.db" | head -1) --skip 2 > $out/kernel_stats_$wl.txt 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d $out/pf_$wl -o pmc -- $K --steps 3 --warmup 1 --repeats 1 > $out/pf_$wl.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE -d $out/pw_$wl -o pmc -- $K --steps 3 --warmup 1 --repeats 1 > $out/pw_$wl.log 2>&1
  python $R/tools/pmc_traffic.py $(find $out/pf_$wl -name "*.db" | head -1) $(find $out/pw_$wl -name "*.db" | head -1) --workload $wl > $out/traffic_$wl.json 2> $out/traffic_$wl.err
  head -12 $out/kernel_stats_$wl.txt
done
# timelines of the pipelined region (two buffer sets in flight through HotPathPipeline)
tl() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace -d $out/tl_$name -o kt -- $B --role kernel --steps 20 --warmup 3 --repeats 6 "$@" > $out/tl_$name.log 2>&1; python $R/tools/overlap_timeline.py $(find $out/tl_$name -name "*.db" | head -1) --tail 0.35 > $out/timeline_$name.txt 2>&1; head -25 $out/timeline_$name.txt; }
tl kfull_hilo --workload kfull
find $out -name "*.db" -delete; find $out -name "*.csv" -size +1M -delete
