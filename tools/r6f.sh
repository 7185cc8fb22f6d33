#!/bin/bash
# round 6, GPU call F: the checkpoint of the FINAL tree -- full `pytest -m gpu`, smoke(), the driver's bench command twice
# (cold processes, wall time), the second-pass leg's hilo / serial comparison with a half-precision model.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; out=$R/gpurun_out/r6f; mkdir -p $out
sha256sum whisper-timestamped_amd/libwtalign.so bench.py benchlib/*.py whisper-timestamped_amd/whisper_timestamped/*.py > $out/sha256_of_what_ran.txt
for i in 1 2; do
  t0=$(date +%s)
  timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_run$i.json 2> $out/bench_run$i.err; echo "{\"run\": $i, \"rc\": $?, \"wall_s\": $(( $(date +%s) - t0 ))}" >> $out/bench_runs.jsonl
  python - <<PY
import json
d=json.loads(open('$out/bench_run$i.json').read().strip().splitlines()[-1])
e=d.get('e2e',{})
print('run $i', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('whole_step'), 'cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline_1thread',{}).get('value'), 'e2e', e.get('audio_s_per_s'), e.get('speedup_vs_cpu_e2e'), (e.get('fp16_model') or {}).get('audio_s_per_s'), 'fail', d.get('parity_failures'), d.get('max_abs_dt_word_vs_ref_s'), [k for k in d if 'error' in k], [k for k in e if 'error' in k], {k: (v.get('ms_per_step'), v.get('error')) for k, v in (d.get('other_configs') or {}).items()})
ds=e.get('default_strategy',{})
print('   default', {k: ds.get(k,{}).get('audio_s_per_s') for k in ('1_stream','32_streams','128_streams','ragged_32_streams','ragged_128_streams')}, ds.get('speedup_vs_cpu'), {k: v.get('audio_s_per_s') for k, v in ds.get('long_form_1h_islands',{}).items() if isinstance(v, dict)})
PY
done
cat $out/bench_runs.jsonl
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "rc=$?" >> $out/smoke.log; tail -2 $out/smoke.log
for v in serial hilo_timeline; do timeout 200 python tools/dbg_batched_hang.py $v 8 2>&1 | grep DONE; done
timeout 2400 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "rc=$?" >> $out/pytest_gpu.log; grep -v "Warning\|warn" $out/pytest_gpu.log | tail -4
