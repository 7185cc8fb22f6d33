#!/bin/bash
# round 6, GPU call Y: the row pass's output stores -- 24 dword stores per row (shipped), six 16-byte stores (st4), the same
# non-temporal (st4nt): kernel traces of the single-stream kfull leg, three processes each, interleaved (one box).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; out=$R/gpurun_out/r6y; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do for v in shipped st4 st4nt; do
  lib=$R/whisper-timestamped_amd/libwtalign.so; [ $v != shipped ] && lib=$R/tools/variants/libwtalign_$v.so
  WT_LIBWTALIGN=$lib timeout 300 rocprofv3 --kernel-trace --stats -d $out/kt_${v}_$rep -o kt -- python $R/bench.py --workload kfull --role kernel --pipeline 1 --steps 10 --warmup 2 --repeats 3 > $out/kt_${v}_$rep.log 2>&1
  python $R/tools/rocpd_stats.py $(find $out/kt_${v}_$rep -name "*.db" | head -1) --skip 2 > $out/kernel_stats_${v}_$rep.txt 2>&1
  echo "== $v $rep: $(grep rowmean $out/kernel_stats_${v}_$rep.txt | awk '{print $3}') + $(grep colnorm $out/kernel_stats_${v}_$rep.txt | awk '{print $3}') + dtw $(grep dtw_kernel $out/kernel_stats_${v}_$rep.txt | awk '{print $3}')"
done; done
find $out -name "*.db" -delete; find $out -name "*.csv" -size +1M -delete
