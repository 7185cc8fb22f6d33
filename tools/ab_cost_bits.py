#!/usr/bin/env python3
"""Digest of wt_cost_batch's output on seeded batches that take the per-class row launches (more than 4096 token rows),
for comparing two builds of libwtalign.so bit for bit:  WT_LIBWTALIGN=<build> python tools/ab_cost_bits.py > digests.json
Every F class from 2 to 5 (512 < F <= 1500), F % 4 in {0, 1, 2, 3}, 1 / 5 / 8 heads, unsorted and sorted unit orders, pad masks."""
import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "whisper-timestamped_amd"))
from whisper_timestamped import _lib as L   # noqa: E402

DEV = torch.device("cuda", 0)


def batch(seed, shapes, n_heads, sort):
    g = torch.Generator(device="cpu").manual_seed(seed)
    if sort:
        shapes = sorted(shapes, key=lambda tf: (tf[1] + 255) // 256)
    n = len(shapes)
    descs = L.make_descs(n)
    off = 0
    A = 10
    for d, (T, F) in zip(descs, shapes):
        start = int(torch.randint(0, 1500 - F + 1, (1,), generator=g))
        d["qk_offset"] = off
        d["head_stride"] = T * 1500
        d["row_stride"] = 1500
        d["T"], d["F"], d["start_token"] = T, F, start
        d["pad_from"] = -1 if (T + F) % 3 else max(F // 2, 1)
        off += A * T * 1500
    n_cost, _, _ = L.layout_outputs(descs)
    qk = (torch.randn(off, generator=g) * 2.0).to(DEV)
    heads = torch.randperm(A, generator=g)[:n_heads].to(torch.int32).to(DEV)
    cost = torch.full((n_cost + 4,), float("nan"), dtype=torch.float32, device=DEV)
    dd = L.descs_to_device(descs, DEV)
    L.cost_batch(qk, descs, dd, heads, cost)
    torch.cuda.synchronize()
    c = cost.cpu().numpy()
    for d in descs:                                    # (the layout may leave gaps between units: those stay NaN)
        assert np.isfinite(c[int(d['cost_offset']):int(d['cost_offset']) + int(d['T']) * int(d['F'])]).all()
    return hashlib.sha256(c.tobytes()).hexdigest()


def main():
    rng = np.random.RandomState(5)
    out = {"lib": L.LIB_PATH}
    k = 0
    for n_heads in (8, 5, 1, 2):
        for sort in (True, False):
            shapes = []
            for cls in (2, 3, 4, 5, 5, 4, 3, 2, 5, 4):
                for r in range(4):
                    F = int(rng.randint(cls * 256 + 1, min((cls + 1) * 256, 1500) + 1))
                    F = F - (F % 4) + ((r + k) % 4)
                    F = max(cls * 256 + 1, min(F, min((cls + 1) * 256, 1500)))
                    shapes.append((int(rng.randint(100, 225)), F))
            shapes.append((224, 1500))
            shapes.append((224, 1280 + 1))
            shapes.append((200, 1536 - 36))
            out[f"batch{k}_heads{n_heads}_{'sorted' if sort else 'mixed'}"] = batch(100 + k, shapes, n_heads, sort)
            k += 1
    print(json.dumps(out))


if __name__ == "__main__":
    main()
