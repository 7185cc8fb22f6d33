#!/bin/bash
# Run on the GPU box (through gpurun):  tools/profile_round.sh <tag> [workload]
# Produces gpurun_out/<tag>/: bench.json (un-profiled default line, e2e leg and CPU baselines included),
# kernel_stats.txt (rocprofv3 --kernel-trace of the same kernel-level bench command), traffic.json (FETCH_SIZE /
# WRITE_SIZE PMC passes, separate runs, gfx950 correction applied, tagged with the workload),
# sq_counters.txt (SQ instruction / busy / wait / LDS-conflict counters per kernel, two more PMC passes),
# e2e_kernel_stats.txt (kernel trace of the transcribe()-level leg alone).
set -u
tag=${1:-round}
wl=${2:-kfull}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$ROOT/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --workload $wl"
K="$B --role kernel --pipeline 1"   # kernel trace / PMC passes: the kernel-level leg itself (bench.py's child process), one batch in flight, stages back to back (what the stage times and the roofline are measured on)
timeout 600 $B --steps 20 --warmup 5 > "$out/bench.json" 2> "$out/bench.err"
timeout 300 rocprofv3 --kernel-trace --stats -d "$out/kt" -o kt -- $K --steps 10 --warmup 2 --repeats 5 > "$out/kt.log" 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d "$out/pmc_fetch" -o pmc -- $K --steps 3 --warmup 1 --repeats 1 > "$out/pmc_fetch.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d "$out/pmc_write" -o pmc -- $K --steps 3 --warmup 1 --repeats 1 > "$out/pmc_write.log" 2>&1
# SQ counters of the same command, two passes (instruction mix, busy / wait cycles, LDS bank conflicts per kernel)
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY -d "$out/pmc_sq1" -o pmc -- $K --steps 3 --warmup 1 --repeats 1 > "$out/pmc_sq1.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS -d "$out/pmc_sq2" -o pmc -- $K --steps 3 --warmup 1 --repeats 1 > "$out/pmc_sq2.log" 2>&1
python $ROOT/tools/pmc_counters.py $(find "$out/pmc_sq1" "$out/pmc_sq2" -name "*.db") > "$out/sq_counters.txt" 2> "$out/sq_counters.err"
if [ "$wl" = kfull ]; then
  timeout 300 rocprofv3 --kernel-trace --stats -d "$out/kt_e2e" -o kt -- $B --role e2e --leg fp32 --no-cpu-baseline --e2e-steps 3 > "$out/kt_e2e.log" 2>&1
  ke=$(find "$out/kt_e2e" -name "*.db" | head -1)
  python $ROOT/tools/rocpd_stats.py "$ke" > "$out/e2e_kernel_stats.txt" 2>&1
fi
kt=$(find "$out/kt" -name "*.db" | head -1)
pf=$(find "$out/pmc_fetch" -name "*.db" | head -1)
pw=$(find "$out/pmc_write" -name "*.db" | head -1)
python $ROOT/tools/rocpd_stats.py "$kt" --skip 2 > "$out/kernel_stats.txt" 2>&1
python $ROOT/tools/pmc_traffic.py "$pf" "$pw" --workload $wl > "$out/traffic.json" 2> "$out/traffic.err"
find "$out" -name "*.csv" -size +2M -delete

tail -1 "$out/bench.json"; head -14 "$out/kernel_stats.txt"; head -30 "$out/e2e_kernel_stats.txt" 2>/dev/null
# secondary workloads (one line each) and the transcribe()-level traces: plain decoding vs the timestamped data planes
WT_BENCH_FORCE_DIST=1 timeout 300 python $ROOT/bench.py --steps 20 --warmup 5 --e2e off --no-cpu-baseline > "$out/bench_force_dist_rccl_1rank.json" 2>> "$out/bench.err" || true
timeout 300 python $ROOT/tools/bench_transcribe.py base > "$out/timestamp_overhead.json" 2> "$out/timestamp_overhead.err" || true
timeout 300 python $ROOT/tools/bench_transcribe.py small > "$out/timestamp_overhead_small.json" 2>> "$out/timestamp_overhead.err" || true
for v in plain timestamped per_segment; do
  timeout 300 rocprofv3 --kernel-trace --stats -d "$out/kt_tr_$v" -o kt -- python $ROOT/tools/bench_transcribe.py base --only $v > "$out/kt_tr_$v.log" 2>&1
  kd=$(find "$out/kt_tr_$v" -name "*.db" | head -1)
  python $ROOT/tools/rocpd_stats.py "$kd" > "$out/transcribe_${v}_kernel_stats.txt" 2>&1
done
find "$out" -name "*.db" -delete
find "$out" -name "*.csv" -delete
