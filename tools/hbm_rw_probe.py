#!/usr/bin/env python3
"""Write / copy / read bandwidth of the box's GPU with torch's own kernels (1 GiB buffers, median of 20): the scale on
which write-heavy kernels (wt_qk_rows_batch: 148 MB written per launch) should be read.  Prints one JSON line."""
import json
import torch

dev = "cuda:0"
n = 1 << 28                      # 1 GiB of fp32
x = torch.empty(n, device=dev)
y = torch.empty(n, device=dev)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e-3)
    return sorted(ts)[len(ts) // 2]


gib = n * 4
out = {"fill_GBps": round(gib / timed(lambda: x.fill_(1.0)) / 1e9, 1),
       "copy_GBps_read_plus_write": round(2 * gib / timed(lambda: y.copy_(x)) / 1e9, 1),
       "sum_GBps_read": round(gib / timed(lambda: x.sum()) / 1e9, 1)}
print(json.dumps(out))
