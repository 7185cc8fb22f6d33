#!/usr/bin/env python3
"""One-off stress (GPU box): wt_cost_batch against the oracle over random batches -- mixed window lengths (every
rowmean class, both launch groups, grouped and ungrouped unit order), fp32 and fp16 rows, padding masks."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "whisper-timestamped_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import synth  # noqa: E402
import test_gpu_parity as T  # noqa: E402
from whisper_timestamped import _lib  # noqa: E402


def main(rounds=10):
    worst = 0.0
    for seed in range(rounds):
        rng = np.random.RandomState(500 + seed)
        n = int(rng.randint(3, 40))
        heads = sorted(rng.choice(24, size=int(rng.randint(1, 9)), replace=False).tolist())
        shapes = []
        for _ in range(n):
            F = int(rng.choice([rng.randint(1, 40), rng.randint(40, 600), rng.randint(600, 1500), 1500]))
            Tn = int(rng.randint(1, min(F, 60) + 1))
            s = int(rng.randint(0, 1500 - F + 1))
            shapes.append((Tn, s, s + F))
        if seed % 2:                                             # grouped the way AlignmentBatch hands units over
            order = _lib.launch_order([(t, e - s) for t, s, e in shapes])
            shapes = [shapes[i] for i in order]
        half = seed % 3 == 2
        qs = [synth.synth_qk(1000 * seed + k, 24, t, lo=s, hi=e) for k, (t, s, e) in enumerate(shapes)]
        if half:
            qs = [q.astype(np.float16).astype(np.float32) for q in qs]
        pads = [(-1 if rng.rand() < 0.6 else int(rng.randint(0, e - s))) for _, s, e in shapes]   # 0 == no mask
        got = T.run_cost(qs, heads, [(s, e) for _, s, e in shapes], pads, dtype=torch.float16 if half else torch.float32)
        for q, (t, s, e), p, g in zip(qs, shapes, pads, got):
            ref = T.oracle_cost(q, heads, (s, e), p)
            err = np.abs(g.astype(np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30)
            worst = max(worst, err)
            assert err < 2e-6, (seed, t, s, e, p, err)
        print(f"round {seed}: {n} units ({'fp16' if half else 'fp32'}, {'grouped' if seed % 2 else 'any order'}) ok, "
              f"worst rel err so far {worst:.2e}", flush=True)


if __name__ == "__main__":
    main()
