#!/bin/bash
# round 6, GPU call Z: the same ablations on the fp16 row pass (large-v3 shapes, 256 units) and on the fp32 one at 256 units.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; out=$R/gpurun_out/r6z; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for wl in largev3_fp16 kfull256; do for v in shipped abl_both abl_nostore abl_nodma; do
  lib=$R/whisper-timestamped_amd/libwtalign.so; [ $v != shipped ] && lib=$R/tools/variants/libwtalign_$v.so
  WT_LIBWTALIGN=$lib timeout 300 rocprofv3 --kernel-trace --stats -d $out/kt_${wl}_$v -o kt -- python $R/bench.py --workload $wl --role kernel --pipeline 1 --steps 10 --warmup 2 --repeats 3 > $out/kt_${wl}_$v.log 2>&1
  python $R/tools/rocpd_stats.py $(find $out/kt_${wl}_$v -name "*.db" | head -1) --skip 2 > $out/kernel_stats_${wl}_$v.txt 2>&1
  echo "== $wl $v: rowmean $(grep rowmean $out/kernel_stats_${wl}_$v.txt | awk '{print $3}') colnorm $(grep colnorm $out/kernel_stats_${wl}_$v.txt | awk '{print $3}')"
done; done
find $out -name "*.db" -delete; find $out -name "*.csv" -size +1M -delete
