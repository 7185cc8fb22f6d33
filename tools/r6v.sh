#!/bin/bash
# round 6, GPU call V: engine clock and socket power while the kernel-level leg runs (rocm-smi sampled every ~0.1 s):
# the single-stream pass (stages back to back) against the pipelined pass (two buffer sets, hilo).  Question: is what the
# pipelined step loses to the sum of its HBM-bound stages (0.35 vs 0.31 ms) a clock / power effect of VALU-bound and
# HBM-bound kernels sharing the chip?
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; out=$R/gpurun_out/r6v; mkdir -p $out
sample() {  # file
  while true; do
    echo "$(date +%s.%N) $(rocm-smi --showclocks --showpower --csv 2>/dev/null | tr '\n' ' ')" >> $1
    sleep 0.05
  done
}
rocm-smi --showclocks --showpower > $out/idle.txt 2>&1
for mode in single pipelined; do
  p=1; [ $mode = pipelined ] && p=2
  sample $out/smi_$mode.txt & S=$!
  timeout 300 python3 bench.py --role kernel --workload kfull --pipeline $p --steps 20 --warmup 5 --min-seconds 6 --out $out/leg_$mode.json > $out/leg_$mode.log 2>&1
  kill $S; wait $S 2>/dev/null
done
python - <<PY
import json, re, statistics
for mode in ("single", "pipelined"):
    rows = open("$out/smi_%s.txt" % mode).read().strip().splitlines()
    sclk, power = [], []
    for r in rows:
        m = re.findall(r"\((\d+)Mhz\)", r)
        w = re.findall(r",(\d+\.\d+)", r)
        if m: sclk.append(max(int(x) for x in m))
        if w: power.append(max(float(x) for x in w))
    try:
        d = json.load(open("$out/leg_%s.json" % mode))
    except Exception as e:
        d = {"error": repr(e)}
    print(json.dumps({"mode": mode, "samples": len(rows), "ms_per_step": d.get("ms_per_step"), "single_ms": (d.get("single_batch_in_flight") or {}).get("ms_per_step"),
                      "sclk_MHz_max_of_row": {"median": statistics.median(sclk) if sclk else None, "min": min(sclk) if sclk else None, "max": max(sclk) if sclk else None},
                      "power_W": {"median": statistics.median(power) if power else None, "max": max(power) if power else None}}))
PY
head -3 $out/smi_pipelined.txt | cut -c1-600
cat $out/idle.txt | head -30
