#!/bin/bash
# round 6, GPU call P: the persistent stft_mel launch with 2 / 3 / 4 (shipped) workgroups per CU (LDS request enforces the residency)
# under hilo, now that the gather beside it takes a quarter of the VALU slots it used to: kfull, alternating, three times.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; out=$R/gpurun_out/r6p; mkdir -p $out
sha256sum whisper-timestamped_amd/libwtalign.so tools/variants/*.so > $out/sha256_of_what_ran.txt
run() {  # name, lib, args...
  name=$1; lib=$2; shift; shift
  WT_LIBWTALIGN=$lib timeout 400 python3 bench.py --no-cpu-baseline --e2e off --other-configs off --steps 20 --warmup 5 "$@" > $out/$name.json 2> $out/$name.err
  python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
    print(json.dumps({"run": "$name", "ms_per_step": d['ms_per_step'], "single": d['single_batch_in_flight']['ms_per_step'], "logmel": d['stages']['logmel']['ms'], "parity": d['parity_in_leg'].get('ok')}))
except Exception as e:
    print(json.dumps({"run": "$name", "error": repr(e)}))
PY
}
for rep in 1 2 3; do
  run kfull_stft4_$rep $R/whisper-timestamped_amd/libwtalign.so
  run kfull_stft3_$rep $R/tools/variants/libwtalign_stft3.so
  run kfull_stft2_$rep $R/tools/variants/libwtalign_stft2.so
done | tee $out/summary.jsonl
