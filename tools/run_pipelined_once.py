#!/usr/bin/env python3
"""The pipelined region of the kernel-level leg WITHOUT its checks: `steps` submits of a workload through
pipeline.HotPathPipeline (two buffer sets, the workload's auto schedule), then synchronize.  For kernel traces of
ABLATED builds of the library (WT_LIBWTALIGN=...), whose results are wrong by construction and would stop bench.py at
its in-leg parity check.    python tools/run_pipelined_once.py [workload] [steps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-timestamped_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import workloads as WL                                              # noqa: E402
from whisper_timestamped.pipeline import HotPathPipeline, choose_schedule    # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "kfull"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    w = WL.make_workload(dev, WL.WORKLOADS[name], seed=1234)
    sets = [w, WL.twin(w)]
    schedule = choose_schedule("auto", len(w["descs"]), w["batch"].fused_small_units)
    pipe = HotPathPipeline(dev, depth=2, schedule=schedule)
    for k in range(8):
        pipe.submit(sets[k % 2]["batch"])
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    import time
    a = time.perf_counter()
    for k in range(steps):
        pipe.submit(sets[k % 2]["batch"])
    torch.cuda.synchronize()
    print(f"{name} {schedule}: {(time.perf_counter() - a) / steps * 1e3:.4f} ms per step over {steps} steps (no checks)")


if __name__ == "__main__":
    main()
