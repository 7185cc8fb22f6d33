#!/bin/bash
# Run on the GPU box (through gpurun):  tools/profile_round.sh <tag>
# Produces gpurun_out/<tag>/: bench.json (un-profiled), kernel_stats.txt (rocprofv3 --kernel-trace --stats of the same
# bench command), traffic.json (FETCH_SIZE / WRITE_SIZE PMC passes, separate runs, gfx950 correction applied).
set -u
tag=${1:-round}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$ROOT/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py"
timeout 300 $B --steps 20 --warmup 3 > "$out/bench.json" 2> "$out/bench.err"
timeout 300 rocprofv3 --kernel-trace --stats -d "$out/kt" -o kt -- $B --steps 10 --warmup 2 --no-cpu-baseline > "$out/kt.log" 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d "$out/pmc_fetch" -o pmc -- $B --steps 3 --warmup 1 --no-cpu-baseline > "$out/pmc_fetch.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d "$out/pmc_write" -o pmc -- $B --steps 3 --warmup 1 --no-cpu-baseline > "$out/pmc_write.log" 2>&1
kt=$(find "$out/kt" -name "*.db" | head -1)
pf=$(find "$out/pmc_fetch" -name "*.db" | head -1)
pw=$(find "$out/pmc_write" -name "*.db" | head -1)
python $ROOT/tools/rocpd_stats.py "$kt" --skip 2 > "$out/kernel_stats.txt" 2>&1
python $ROOT/tools/pmc_traffic.py "$pf" "$pw" > "$out/traffic.json" 2> "$out/traffic.err"
find "$out" -name "*.csv" -size +2M -delete
tail -1 "$out/bench.json"; head -12 "$out/kernel_stats.txt"; cat "$out/traffic.json"
