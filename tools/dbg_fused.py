#!/usr/bin/env python3
"""Where do the fused small-unit kernel's cost matrices differ from the batched kernels'?  (debug helper, GPU box)"""
import sys, os
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R, "whisper-timestamped_amd"), os.path.join(R, "tests")):
    sys.path.insert(0, p)
import pytest  # noqa
import test_gpu_parity as P
L = P._lib()
for shapes in ([(64, 0, 256)], [(11, 3, 147)], [(5, 1, 8)], [(30, 0, 1025)], [(16, 0, 1500)], [(64, 0, 256), (11, 3, 147), (63, 7, 300)]):
    for dtype in (torch.float32,):
        a = P._align_units(shapes, dtype, L.WT_ALIGN_KEEP_COST)
        b = P._align_units(shapes, dtype, L.WT_ALIGN_KEEP_COST | L.WT_ALIGN_NO_FUSED_SMALL_UNITS)
        for d in a["descs"]:
            T, F, c0 = int(d["T"]), int(d["F"]), int(d["cost_offset"])
            x = a["cost"][c0:c0 + T * F].reshape(T, F).cpu().numpy()
            y = b["cost"][c0:c0 + T * F].reshape(T, F).cpu().numpy()
            bad = np.argwhere(~((x == y) | (np.isnan(x) & np.isnan(y))))
            j0 = int(d["jumps_offset"])
            jeq = bool(torch.equal(a["jumps"][j0:j0 + T + 1], b["jumps"][j0:j0 + T + 1]))
            print(f"shapes={shapes} unit T={T} F={F} pad={int(d['pad_from'])}: {len(bad)} cells differ, jumps equal: {jeq}")
            if len(bad):
                rows, cols = np.unique(bad[:, 0]), np.unique(bad[:, 1])
                print("   rows:", rows[:12], "... cols:", cols[:12], "... n_rows", len(rows), "n_cols", len(cols))
                for (t, f) in bad[:6]:
                    print(f"   [{t},{f}] fused={x[t, f]!r} batched={y[t, f]!r} rel={(x[t,f]-y[t,f])/abs(y[t,f]) if y[t,f] else 0:.3e}")
