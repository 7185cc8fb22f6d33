#!/bin/bash
# round 6, GPU call AC: why does dtw_kernel take 230 us beside the gathers (97 alone)?  Ablation: the sweep without its
# direction-plane stores (results wrong; timing only), in the pipelined region (two buffer sets, hilo) and alone; kernel traces.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; out=$R/gpurun_out/r6ac; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in shipped abl_dtw_nostore; do
  lib=$R/whisper-timestamped_amd/libwtalign.so; [ $v != shipped ] && lib=$R/tools/variants/libwtalign_$v.so
  WT_LIBWTALIGN=$lib timeout 300 rocprofv3 --kernel-trace --stats -d $out/kt_$v -o kt -- python $R/tools/run_pipelined_once.py kfull 200 > $out/kt_$v.log 2>&1
  tail -1 $out/kt_$v.log
  python $R/tools/rocpd_stats.py $(find $out/kt_$v -name "*.db" | head -1) --skip 2 > $out/kernel_stats_pipelined_$v.txt 2>&1
  echo "== pipelined $v"; grep "dtw_kernel\|logprob_gather\|stft_mel\|rowmean" $out/kernel_stats_pipelined_$v.txt | cut -c1-130
  python $R/tools/gantt.py $(find $out/kt_$v -name "*.db" | head -1) --steps 4 > $out/gantt_$v.txt 2>&1
done
find $out -name "*.db" -delete; find $out -name "*.csv" -size +1M -delete
