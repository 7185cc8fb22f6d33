#!/usr/bin/env python3
"""One-off stress (GPU box): bit-exactness of wt_dtw_batch against the oracle over many random shapes / seeds /
cost structures (uniform, quantised with exact ties, ridge + mask).  Not part of the test suite (minutes of oracle time)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "whisper-timestamped_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import test_gpu_parity as T  # noqa: E402


def main(n_rounds=12, per_round=120):
    total = 0
    for seed in range(100, 100 + n_rounds):
        rng = np.random.RandomState(seed)
        costs = []
        for _ in range(per_round):
            Tn = int(rng.randint(1, 257))
            Fn = int(rng.randint(1, 1793))
            if Tn > 192 and Fn > 1760:
                Fn = 1760
            kind = rng.randint(4)
            if kind == 0:
                c = -rng.rand(Tn, Fn)
            elif kind == 1:
                c = -(rng.randint(0, 3, size=(Tn, Fn)) / 2.0)
            elif kind == 2:
                c = -rng.rand(Tn, Fn) * 1e-3
                st = np.sort(rng.randint(0, Fn, size=Tn))
                c[np.arange(Tn), st] -= 0.5
                c[:-1, int(Fn * rng.rand()):] = 0.0
            else:
                c = -np.abs(rng.standard_normal((Tn, Fn))) * 10.0 ** rng.uniform(-12, 0, size=(Tn, 1))
            c = c.astype(np.float32)
            c[0, 0] = c.min()
            costs.append(c)
        T.check_dtw_exact(costs)
        total += len(costs)
        print(f"seed {seed}: {len(costs)} units bit-exact (cumulative {total})", flush=True)


if __name__ == "__main__":
    main()
