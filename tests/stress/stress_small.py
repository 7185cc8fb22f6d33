#!/usr/bin/env python3
"""One-off stress (GPU box): the fused tail kernel of wt_align_batch_v3 (wt_small.hip) against the batched kernels and the
oracle over many random small units -- every shape that qualifies (one to three sweeping waves, F from 1 to 1792),
fp32 and fp16 rows, with and without pad masks: cost, jumps, path and distance bit-identical to the batched kernels',
jumps bit-exact against the oracle DTW run on that cost.  Not part of the test suite."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "whisper-timestamped_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import test_gpu_parity as T  # noqa: E402
from oracle import align_ref as O  # noqa: E402


def main(n_rounds=8, per_round=90):
    L = T._lib()
    total = waves = 0
    for seed in range(n_rounds):
        rng = np.random.RandomState(700 + seed)
        shapes = []
        while len(shapes) < per_round:
            Tn = int(np.clip(np.round(np.exp(rng.normal(np.log(14), 1.0))), 1, 200))
            Fn = int(np.clip(np.round(np.exp(rng.normal(np.log(150), 0.9))), 1, 1500))
            if not L.small_unit(Tn, Fn):
                continue
            s = int(rng.randint(0, 1500 - Fn + 1))
            shapes.append((Tn, s, s + Fn))
        dtype = torch.float16 if seed % 2 else torch.float32
        a = T._align_units(shapes, dtype, L.WT_ALIGN_KEEP_COST, seed=4000 + 97 * seed, n_heads=6)
        b = T._align_units(shapes, dtype, L.WT_ALIGN_KEEP_COST | L.WT_ALIGN_NO_FUSED_SMALL_UNITS, seed=4000 + 97 * seed, n_heads=6)
        for k, d in enumerate(a["descs"]):
            Tn, Fn = int(d["T"]), int(d["F"])
            c0, j0, p0 = int(d["cost_offset"]), int(d["jumps_offset"]), int(d["path_offset"])
            ca = a["cost"][c0:c0 + Tn * Fn]
            assert torch.equal(ca, b["cost"][c0:c0 + Tn * Fn]), (Tn, Fn)
            assert torch.equal(a["jumps"][j0:j0 + Tn + 1], b["jumps"][j0:j0 + Tn + 1]), (Tn, Fn)
            n = int(b["pl"][k])
            assert int(a["pl"][k]) == n and torch.equal(a["pi"][p0:p0 + n], b["pi"][p0:p0 + n]) and torch.equal(a["pj"][p0:p0 + n], b["pj"][p0:p0 + n])
            assert float(a["dist"][k]) == float(b["dist"][k])
            r = O.dtw_ref(ca.reshape(Tn, Fn).cpu().numpy().astype(np.float64))
            assert np.array_equal(a["jumps"][j0:j0 + Tn + 1].cpu().numpy(), O.jumps_from_path(r.index1s, r.index2s)), (Tn, Fn)
            total += 1
            waves += int(Tn > 64)
    print(f"stress_small: {total} units ({waves} of them with more than one sweeping wave), fused == batched bit for bit, "
          f"jumps == oracle DTW of the same cost")


if __name__ == "__main__":
    main()
