#!/usr/bin/env python3
"""A/B of two builds of libwtalign.so on the cost stage (GPU box): each library (WT_LIBWTALIGN) in its own subprocess,
alternating, same box: microseconds of wt_cost_batch on the K-full batch (32 units of 8 x 224 x 1500) and the worst
relative error against the oracle over tests/stress/stress_cost.py's random batches.
usage: ab_cost.py <libA.so> <libB.so>"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path[:0] = [ROOT, os.path.join(ROOT, "whisper-timestamped_amd"), os.path.join(ROOT, "tests")]
    import numpy as np
    import torch
    import bench
    from whisper_timestamped import _lib
    dev = torch.device("cuda", 0)
    w = bench.make_workload(dev, dict(bench.WORKLOADS["kfull"]), seed=1234)
    calls = bench._stage_calls(w)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        calls["cost"](st)
    torch.cuda.synchronize()
    times = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(100):
            calls["cost"](st)
        b.record()
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b) / 100 * 1e3)
    # accuracy on random batches (the stress tool's generator, 4 rounds)
    import io
    import contextlib
    sys.path.insert(0, os.path.join(ROOT, "tests", "stress"))
    import stress_cost
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        stress_cost.main(4)
    worst = buf.getvalue().strip().splitlines()[-1].split("worst rel err so far")[-1].strip()
    print(json.dumps(dict(lib=os.path.basename(_lib.LIB_PATH), cost_stage_us_median=round(sorted(times)[3], 2),
                          us_min=round(min(times), 2), worst_rel_err_vs_oracle=worst)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 2 and sys.argv[1] == "child":
        child()
    else:
        libs = [os.path.abspath(p) for p in sys.argv[1:3]]
        for lib in libs + libs:
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, WT_LIBWTALIGN=lib), check=True)
