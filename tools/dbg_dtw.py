#!/usr/bin/env python3
"""Debug helper (GPU box): run wt_dtw_batch on named shape sets in subprocesses, report which ones fault / mismatch."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def run_raw(shapes, path, dist):
    import numpy as np, torch
    sys.path[:0] = [ROOT, os.path.join(ROOT, "whisper-timestamped_amd"), os.path.join(ROOT, "tests")]
    from whisper_timestamped import _lib as L
    from oracle import align_ref as O
    rng = np.random.RandomState(0)
    costs = [(-rng.rand(T, F)).astype(np.float32) for T, F in shapes]
    descs = L.make_descs(len(costs))
    for d, c in zip(descs, costs):
        d["T"], d["F"] = c.shape
        d["pad_from"] = -1
    n_cost, n_jumps, n_path = L.layout_outputs(descs)
    flat = np.zeros(n_cost, dtype=np.float32)
    for d, c in zip(descs, costs):
        flat[d["cost_offset"]:d["cost_offset"] + c.size] = c.ravel()
    dev = "cuda:0"
    jumps = torch.full((n_jumps,), -7, dtype=torch.int32, device=dev)
    pi = torch.full((n_path + 64,), -7, dtype=torch.int32, device=dev) if path else None
    pj = torch.full((n_path + 64,), -7, dtype=torch.int32, device=dev) if path else None
    pl = torch.zeros(len(costs), dtype=torch.int32, device=dev) if path else None
    ds = torch.zeros(len(costs), dtype=torch.float64, device=dev) if dist else None
    L.dtw_batch(torch.from_numpy(flat).to(dev), descs, L.descs_to_device(descs, dev), jumps, pi, pj, pl, ds)
    torch.cuda.synchronize()
    j = jumps.cpu().numpy()
    for k, (d, c) in enumerate(zip(descs, costs)):
        r = O.dtw_ref(c.astype(np.float64))
        assert np.array_equal(j[d["jumps_offset"]:d["jumps_offset"] + c.shape[0] + 1], O.jumps_from_path(r.index1s, r.index2s)), ("jumps", c.shape)
        if path:
            n = int(pl[k])
            assert n == len(r.index1s), ("len", c.shape, n, len(r.index1s))
            assert np.array_equal(pi.cpu().numpy()[d["path_offset"]:d["path_offset"] + n], r.index1s), ("path", c.shape)
        if dist:
            assert float(ds[k]) == r.distance, ("dist", c.shape)


RAW = {
    "t11_dist": ([(1, 1)], False, True), "t11_path": ([(1, 1)], True, False),
    "t22_dist": ([(2, 2)], False, True), "t22_path": ([(2, 2)], True, False),
    "t37_path": ([(3, 7)], True, False), "t2561_path": ([(256, 1)], True, False), "t2561_dist": ([(256, 1)], False, True),
    "t115_path": ([(1, 15)], True, False), "t115_both": ([(1, 15)], True, True),
}
CASES = {
    "kfull_nopath": ([(224, 1500)] * 3, False),
    "kfull_path": ([(224, 1500)] * 3, True),
    "one_wave_path": ([(33, 144), (64, 352), (7, 100)], True),
    "two_waves_path": ([(100, 352), (128, 700)], True),
    "tiny_nopath": ([(1, 1), (2, 2), (3, 7), (256, 1)], False),
    "tiny_path": ([(1, 1), (2, 2), (3, 7), (256, 1)], True),
    "mixed_path": ([(1, 15), (16, 48), (65, 100), (129, 352), (200, 700), (256, 1500)], True),
    "small_F": ([(64, 5), (65, 16), (128, 49)], True),
}
def child(name):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "whisper-timestamped_amd"), os.path.join(ROOT, "tests")]
    import numpy as np
    import test_gpu_parity as P
    from oracle import align_ref as O
    shapes, want_path = CASES[name]
    rng = np.random.RandomState(0)
    costs = [(-rng.rand(T, F)).astype(np.float32) for T, F in shapes]
    if want_path:
        P.check_dtw_exact(costs)
    else:
        import torch
        L = P._lib()
        descs = L.make_descs(len(costs))
        for d, c in zip(descs, costs):
            d["T"], d["F"] = c.shape
            d["pad_from"] = -1
        n_cost, n_jumps, n_path = L.layout_outputs(descs)
        flat = np.zeros(n_cost, dtype=np.float32)
        for d, c in zip(descs, costs):
            flat[d["cost_offset"]:d["cost_offset"] + c.size] = c.ravel()
        jumps = torch.full((n_jumps,), -7, dtype=torch.int32, device="cuda:0")
        L.dtw_batch(torch.from_numpy(flat).cuda(), descs, L.descs_to_device(descs, "cuda:0"), jumps)
        torch.cuda.synchronize()
        j = jumps.cpu().numpy()
        for d, c in zip(descs, costs):
            r = O.dtw_ref(c.astype(np.float64))
            assert np.array_equal(j[d["jumps_offset"]:d["jumps_offset"] + c.shape[0] + 1], O.jumps_from_path(r.index1s, r.index2s))
    print(name, "ok")
if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] in RAW:
        run_raw(*RAW[sys.argv[1]])
        print(sys.argv[1], "ok")
    elif len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for name in list(RAW) + ["tiny_path"]:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), name], capture_output=True, text=True, timeout=120)
            tail = (r.stdout + r.stderr).strip().splitlines()
            print(name, "rc", r.returncode, "|", [l for l in tail if "ok" in l or "Error" in l or "fault" in l or "assert" in l.lower()][-2:])
