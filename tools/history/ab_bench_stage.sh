#!/bin/bash
# A/B of two builds of the library on the whole kernel-level step (GPU box): whisper-timestamped_amd/libwtalign_prev.so
# against the in-tree library, alternating, single stream (per-stage times) -- prints stage ms, step ms, and (with
# "pmc" as first argument) the HBM bytes per launch of every kernel for the in-tree build.  Writes gpurun_out/ab_stage/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/ab_stage; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
for l in prev new prev new; do
  if [ $l = prev ]; then export WT_LIBWTALIGN=$R/whisper-timestamped_amd/libwtalign_prev.so; else unset WT_LIBWTALIGN; fi
  timeout 200 python $R/bench.py --e2e off --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'lib':'$l','ms_per_step':d['ms_per_step'],'single':d['single_batch_in_flight']['ms_per_step'],'stages_ms':{k:v['ms'] for k,v in d['stages'].items()}}))"
done | tee $out/ab.jsonl
unset WT_LIBWTALIGN
if [ "${1:-}" = pmc ]; then
  K="python $R/bench.py --e2e off --no-cpu-baseline --pipeline 1 --steps 3 --warmup 1 --repeats 1"
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d $out/f -o pmc -- $K > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE -d $out/w -o pmc -- $K > /dev/null 2>&1
  python $R/tools/pmc_traffic.py $(find $out/f -name "*.db" | head -1) $(find $out/w -name "*.db" | head -1) --workload kfull > $out/traffic.json
  python -c "
import json
d=json.load(open('$out/traffic.json'))
print({k.split('wt')[1][2:18]:(v['fetch_bytes']//1000000, v['write_bytes']//1000000) for k,v in d.items() if isinstance(v,dict)})"
  find $out -name "*.db" -delete; find $out -name "*.csv" -delete
fi
