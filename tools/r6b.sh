#!/bin/bash
# round 6, GPU call B: parity on the PEAKED double (goldens, B streams vs one stream, GPU vs the CPU reference path), the N4 GPU
# tests (checkpoint files, own tokenizer, CLI), the DTW kernel against the step-pattern interpreter, the pipeline tests again,
# and bench.py's default-strategy leg on its own.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; out=$R/gpurun_out/r6b; mkdir -p $out
run() { name=$1; shift; timeout 1500 python -m pytest "$@" -m gpu -q -s > $out/$name.log 2>&1; echo "rc=$?" >> $out/$name.log; grep -v "Warning\|warn" $out/$name.log | tail -6; }
run pytest_streams tests/test_gpu_streams.py
run pytest_n4 tests/test_gpu_checkpoint_cli.py
run pytest_dtw_interpreter tests/test_gpu_parity.py -k "interpreter or ties"
run pytest_peaked_goldens tests/test_gpu_transcribe.py -k "peaked"
run pytest_streams_batch tests/test_gpu_streams_batch.py -k "ragged or peaked"
timeout 900 python bench.py --role e2e --leg efficient --out $out/default_strategy_leg.json > $out/default_strategy_leg.log 2>&1; echo "leg rc=$?"
python - <<PY
import json
d=json.load(open('$out/default_strategy_leg.json'))
for k in ('1_stream','32_streams','128_streams','ragged_32_streams','ragged_128_streams'):
    print(k, d[k]['audio_s_per_s'], d[k].get('parity_vs_1_stream'))
for k,v in d['long_form_1h_islands'].items():
    if isinstance(v,dict): print('long', k, v['audio_s_per_s'], v.get('parity_vs_1_stream'))
print('flat', d.get('flat_attention_ragged_streams_vs_1_stream'))
p=d.get('parity_vs_cpu_reference_path',{})
print('cpu path', {k:v for k,v in p.items() if k!='recordings_of_the_timed_batches'})
print('timed', p.get('recordings_of_the_timed_batches'))
print('cpu_baseline', d.get('cpu_baseline'), d.get('speedup_vs_cpu'))
print('failures', d.get('parity_failures'))
PY
