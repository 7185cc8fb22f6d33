#!/bin/bash
# round 6, GPU call L: Gantt of the pipelined region (kfull, hilo): where does the step spend the time above the sum of its HBM-bound kernels?
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; out=$R/gpurun_out/r6l; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
timeout 300 rocprofv3 --kernel-trace -d $out/tl -o kt -- $B --role kernel --steps 20 --warmup 3 --repeats 6 --workload kfull > $out/tl.log 2>&1
python $R/tools/gantt.py $(find $out/tl -name "*.db" | head -1) --steps 5 > $out/gantt_kfull_hilo.txt 2>&1
cat $out/gantt_kfull_hilo.txt
find $out -name "*.db" -delete; find $out -name "*.csv" -size +1M -delete
