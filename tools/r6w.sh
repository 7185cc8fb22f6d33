#!/bin/bash
# round 6, GPU call W: what bounds rowmean<24,f32>?  Ablations of its VALU stream (results wrong by construction; timing only):
# the quarter-rate v_exp_f32 replaced by a packed multiply-add, the median network by the centre element, both.
# Kernel trace of the single-stream kfull leg (256 units too), one box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; out=$R/gpurun_out/r6w2; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for wl in kfull; do
for v in shipped abl_nostore abl_nont abl_dmaonly abl_nodma abl_both; do
  lib=$R/whisper-timestamped_amd/libwtalign.so; [ $v != shipped ] && lib=$R/tools/variants/libwtalign_$v.so
  WT_LIBWTALIGN=$lib timeout 300 rocprofv3 --kernel-trace --stats -d $out/kt_${wl}_$v -o kt -- python $R/bench.py --workload $wl --role kernel --pipeline 1 --steps 10 --warmup 2 --repeats 3 > $out/kt_${wl}_$v.log 2>&1
  python $R/tools/rocpd_stats.py $(find $out/kt_${wl}_$v -name "*.db" | head -1) --skip 2 > $out/kernel_stats_${wl}_$v.txt 2>&1
  echo "== $wl $v"; grep "rowmean\|colnorm" $out/kernel_stats_${wl}_$v.txt | cut -c1-140
done; done
find $out -name "*.db" -delete; find $out -name "*.csv" -size +1M -delete
