#!/bin/bash
# Round-4 profile set (GPU box, through gpurun):  tools/profile_r4.sh <tag>  -> gpurun_out/<tag>/
#   bench.json                      the driver's command (python bench.py --gpus 1 --steps 20 --warmup 5), un-profiled
#   kernel_stats_<wl>.txt           rocprofv3 --kernel-trace of the kernel-level leg of every single-GPU BASELINE workload
#   traffic_<wl>.json               FETCH_SIZE / WRITE_SIZE PMC passes (separate runs; gfx950 correction in pmc_traffic.py)
#   sq_counters_<wl>.txt            SQ instruction / busy / wait counters (two more PMC passes) for kfull and largev3_fp16
set -u
tag=${1:-r4}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$ROOT/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?"
for wl in kfull kreal kfull256 largev3_fp16; do
  K="python $ROOT/bench.py --workload $wl --role kernel --pipeline 1"
  timeout 300 rocprofv3 --kernel-trace --stats -d "$out/kt_$wl" -o kt -- $K --steps 10 --warmup 2 --repeats 5 > "$out/kt_$wl.log" 2>&1
  python $ROOT/tools/rocpd_stats.py "$(find "$out/kt_$wl" -name "*.db" | head -1)" --skip 2 > "$out/kernel_stats_$wl.txt" 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d "$out/pf_$wl" -o pmc -- $K --steps 3 --warmup 1 --repeats 1 > "$out/pf_$wl.log" 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE -d "$out/pw_$wl" -o pmc -- $K --steps 3 --warmup 1 --repeats 1 > "$out/pw_$wl.log" 2>&1
  python $ROOT/tools/pmc_traffic.py "$(find "$out/pf_$wl" -name "*.db" | head -1)" "$(find "$out/pw_$wl" -name "*.db" | head -1)" --workload $wl > "$out/traffic_$wl.json" 2> "$out/traffic_$wl.err"
  if [ "$wl" = kfull ] || [ "$wl" = largev3_fp16 ]; then
    timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY -d "$out/sq1_$wl" -o pmc -- $K --steps 3 --warmup 1 --repeats 1 > "$out/sq1_$wl.log" 2>&1
    timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS -d "$out/sq2_$wl" -o pmc -- $K --steps 3 --warmup 1 --repeats 1 > "$out/sq2_$wl.log" 2>&1
    python $ROOT/tools/pmc_counters.py $(find "$out/sq1_$wl" "$out/sq2_$wl" -name "*.db") > "$out/sq_counters_$wl.txt" 2> "$out/sq_counters_$wl.err"
  fi
  head -12 "$out/kernel_stats_$wl.txt"
done
find "$out" -name "*.db" -delete
find "$out" -name "*.csv" -delete
tail -c 3000 "$out/bench.json"
