#!/bin/bash
# round 6, GPU call H: does the DTW launch of 256 units run faster when no CU can take two units (WT_DTW_SPREAD=1: > 80 KB of
# LDS per workgroup for launches of (CUs/2, CUs] units)?  kfull256 / largev3_fp16 kernel legs, both settings, twice.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; out=$R/gpurun_out/r6h; mkdir -p $out
sha256sum whisper-timestamped_amd/libwtalign.so > $out/sha256_of_what_ran.txt
for rep in 1 2; do for wl in kfull256 largev3_fp16; do for sp in 0 1; do
  WT_DTW_SPREAD=$sp timeout 400 python3 bench.py --workload $wl --no-cpu-baseline --e2e off --other-configs off --steps 20 --warmup 5 > $out/${wl}_spread${sp}_$rep.json 2> $out/${wl}_spread${sp}_$rep.err
  python - <<PY
import json
try:
    d=json.loads(open('$out/${wl}_spread${sp}_$rep.json').read().strip().splitlines()[-1])
    print(json.dumps({"workload": "$wl", "spread": $sp, "rep": $rep, "ms_per_step": d['ms_per_step'], "single": d['single_batch_in_flight']['ms_per_step'], "stages": {k: v['ms'] for k, v in d['stages'].items()}, "parity": d['parity_in_leg'].get('ok')}))
except Exception as e:
    print(json.dumps({"workload": "$wl", "spread": $sp, "rep": $rep, "error": repr(e)}))
PY
done; done; done | tee $out/summary.jsonl
