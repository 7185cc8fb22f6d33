#!/bin/bash
# Final freeze of round 3 (GPU box): tests + smoke + the driver's command x3 (tools/gpu_checkpoint.sh), then the kernel trace and
# the PMC traffic of the kernel-level leg, the secondary workloads, and the kernel trace of the real-shape batch.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-r3i}
bash $R/tools/gpu_checkpoint.sh $tag 3
out=$R/gpurun_out/$tag; cd /tmp; export TMPDIR=/tmp
K="python $R/bench.py --role kernel --pipeline 1"
timeout 300 rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- $K --steps 10 --warmup 2 --repeats 5 > $out/kt.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $out/pf -o pmc -- $K --steps 3 --warmup 1 --repeats 1 > $out/pf.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $out/pw -o pmc -- $K --steps 3 --warmup 1 --repeats 1 > $out/pw.log 2>&1
python $R/tools/rocpd_stats.py $(find $out/kt -name "*.db" | head -1) --skip 2 > $out/kernel_stats.txt 2>&1
python $R/tools/pmc_traffic.py $(find $out/pf -name "*.db" | head -1) $(find $out/pw -name "*.db" | head -1) --workload kfull > $out/traffic.json 2> $out/traffic.err
for a in fused split; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $out/kt_kreal_$a -o kt -- $K --workload kreal --align $a --steps 20 --warmup 3 --repeats 2 > /dev/null 2>&1
  python $R/tools/rocpd_stats.py $(find $out/kt_kreal_$a -name "*.db" | head -1) --skip 3 | grep -v at6native > $out/kreal_${a}_kernel_stats.txt 2>&1
done
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $out/pf_kreal -o pmc -- $K --workload kreal --align fused --steps 3 --warmup 1 --repeats 1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $out/pw_kreal -o pmc -- $K --workload kreal --align fused --steps 3 --warmup 1 --repeats 1 > /dev/null 2>&1
python $R/tools/pmc_traffic.py $(find $out/pf_kreal -name "*.db" | head -1) $(find $out/pw_kreal -name "*.db" | head -1) --workload kreal > $out/traffic_kreal.json 2>> $out/traffic.err
cd $R
for w2 in kreal kfull256 largev3_fp16; do
  timeout 300 python bench.py --workload $w2 --steps 10 --warmup 3 --e2e off --no-cpu-baseline > $out/bench_$w2.json 2> $out/bench_$w2.err || true
done
timeout 300 python bench.py --workload kreal --align split --steps 10 --warmup 3 --e2e off --no-cpu-baseline > $out/bench_kreal_split.json 2>> $out/bench_kreal.err || true
find $out -name "*.db" -delete; find $out -name "*.csv" -delete
head -12 $out/kernel_stats.txt | cut -c1-150; head -8 $out/kreal_fused_kernel_stats.txt | cut -c1-150
