#!/usr/bin/env python3
"""Does the hilo schedule's speed depend on WHICH HIP streams it got?  (Fresh-process runs of the same command are bimodal:
most at 0.365 ms per step, one in six near 0.40.)  One process, S stream sets created one after the other, each timed for
R regions of 20 steps, twice round-robin: if a set is consistently slow, the mapping of streams to hardware queues is the
cause and a short calibration before the timed region can pick a good set."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
w = bench.make_workload(dev, bench.WORKLOADS["kfull"], seed=1234)
w["align"] = "split"
P, S, R, steps = 2, int(sys.argv[1]) if len(sys.argv) > 1 else 5, 12, 20
pipe = [w]
for _ in range(P - 1):
    c = dict(w)
    c.pop("_calls", None)
    c.update(cost=torch.empty(w["cost"].numel(), dtype=torch.float32, device=dev), mel=torch.empty_like(w["mel"]),
             gmax=torch.empty_like(w["gmax"]), pad=torch.empty_like(w["pad"]),
             **bench.result_buffers(w["jumps"].numel(), w["logprob"].numel(), dev))
    pipe.append(c)
plan = bench.SCHEDULES["hilo"]
sets = []
for s in range(S):
    shared = {}
    sets.append([bench.plan_streams(dev, plan, shared) for _ in range(P)])


def region(ss):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        bench.run_step_plan(pipe[k % P], plan, ss[k % P])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for ss in sets:
    for _ in range(3):
        region(ss)
out = []
for rnd in range(2):
    for i, ss in enumerate(sets):
        r = [region(ss) for _ in range(R)]
        rec = {"round": rnd, "stream_set": i, "ms_per_step_median": round(float(np.median(r)), 4), "min": round(min(r), 4), "max": round(max(r), 4)}
        print(json.dumps(rec), flush=True)
        out.append(rec)
