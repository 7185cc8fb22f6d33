#!/bin/bash
# round 6, last GPU call: the tree as committed at the end of the round (library unchanged since r6k):
# sha256 of what ran, `pytest -m gpu`, smoke(), the driver's bench command twice from cold processes.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; out=$R/gpurun_out/r6final; mkdir -p $out
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
