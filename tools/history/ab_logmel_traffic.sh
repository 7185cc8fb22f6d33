#!/bin/bash
# A/B of two builds of the library on the log-mel stage (GPU box): whisper-timestamped_amd/libwtalign_prev.so against the
# in-tree library -- time and output checksum (tools/ab_logmel.py), then HBM bytes per launch of both builds
# (rocprofv3 FETCH_SIZE / WRITE_SIZE passes of tools/run_logmel_once.py through tools/pmc_traffic.py).
# Writes gpurun_out/r2n/.  Used for the XCD-aware tile walk of stft_mel (profiles/r2n_*).
mkdir -p gpurun_out/r2n; R=$PWD; cd /tmp; export TMPDIR=/tmp
for l in prev new prev new; do if [ $l = prev ]; then export WT_LIBWTALIGN=$R/whisper-timestamped_amd/libwtalign_prev.so; else unset WT_LIBWTALIGN; fi; WT_AB_LABEL=$l timeout 120 python $R/tools/ab_logmel.py child; done > $R/gpurun_out/r2n/ab_logmel_xcd.jsonl 2>/dev/null
grep -v "\"n_chunks\": 1," $R/gpurun_out/r2n/ab_logmel_xcd.jsonl
for l in prev new; do if [ $l = prev ]; then export WT_LIBWTALIGN=$R/whisper-timestamped_amd/libwtalign_prev.so; else unset WT_LIBWTALIGN; fi
  timeout 200 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/r2n/f_$l -o pmc -- python $R/tools/run_logmel_once.py 32 > /dev/null 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/r2n/w_$l -o pmc -- python $R/tools/run_logmel_once.py 32 > /dev/null 2>&1
  python $R/tools/pmc_traffic.py $(find $R/gpurun_out/r2n/f_$l -name "*.db" | head -1) $(find $R/gpurun_out/r2n/w_$l -name "*.db" | head -1) > $R/gpurun_out/r2n/traffic_logmel_$l.json
  python -c "
import json,sys
d=json.load(open('$R/gpurun_out/r2n/traffic_logmel_$l.json'))
print('$l', {k.split('wt')[1][:22]:(v['fetch_bytes']//1000000, v['write_bytes']//1000000) for k,v in d.items() if isinstance(v,dict)})"
done
unset WT_LIBWTALIGN; cd $R; find gpurun_out/r2n -name "*.db" -delete; find gpurun_out/r2n -name "*.csv" -delete
timeout 200 python -m pytest tests/test_gpu_parity.py -q -x -k logmel 2>&1 | tail -1
