#!/usr/bin/env python3
"""How should the stages of the K-full step share the chip?  (0.405 ms per step against a 0.33 ms
HBM floor; only the DTW overlaps.)  Times the same step under a matrix of schedules, all on the CURRENT kernels:

  batches in flight P            1 .. 4 (each with its own output buffers, its own HIP stream(s))
  stage order per batch          same | staggered (batch j starts with the log-prob gather / with the cost stage)
  stream priorities              none | hilo: per batch a HIGH-priority stream for the latency / VALU-bound kernels
                                 (stft_mel, dtw_kernel) and a LOW-priority one for the HBM-bound ones (cost, log-prob)

    python tools/overlap_matrix.py [--steps 20] [--regions 25] [--out gpurun_out/overlap.json]

Every variant's results are checked against the first buffer set of the plain schedule (bit-identical records)."""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ORDERS = {
    "same": [["logmel", "cost", "dtw", "logprob"]] * 4,
    "stagger_lp": [["logmel", "cost", "dtw", "logprob"], ["logprob", "logmel", "cost", "dtw"],
                   ["cost", "dtw", "logprob", "logmel"], ["logprob", "cost", "dtw", "logmel"]],
    "stagger_cost": [["logmel", "cost", "dtw", "logprob"], ["cost", "dtw", "logprob", "logmel"],
                     ["logprob", "logmel", "cost", "dtw"], ["cost", "logprob", "dtw", "logmel"]],
    "dtw_last": [["logprob", "logmel", "cost", "dtw"]] * 4,
    "lp_first_alt": [["logprob", "logmel", "cost", "dtw"], ["logmel", "cost", "dtw", "logprob"]] * 2,
}


def priority_range():
    hip = ctypes.CDLL("libamdhip64.so")
    lo, hi = ctypes.c_int(), ctypes.c_int()
    rc = hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi))
    return rc, lo.value, hi.value


def make_stream(dev, priority=None):
    if priority is None:
        return torch.cuda.Stream(device=dev)
    hip = ctypes.CDLL("libamdhip64.so")
    st = ctypes.c_void_p()
    rc = hip.hipStreamCreateWithPriority(ctypes.byref(st), ctypes.c_uint(1), ctypes.c_int(priority))     # 1 = hipStreamNonBlocking
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)


def buffer_sets(w, dev, n):
    sets = [w]
    for _ in range(n - 1):
        c = dict(w)
        c.pop("_calls", None)
        c.update(cost=torch.empty(w["cost"].numel(), dtype=torch.float32, device=dev), mel=torch.empty_like(w["mel"]),
                 gmax=torch.empty_like(w["gmax"]), pad=torch.empty_like(w["pad"]),
                 **bench.result_buffers(w["jumps"].numel(), w["logprob"].numel(), dev))
        sets.append(c)
    return sets


def step_serial(c, order, st):
    calls = c.setdefault("_calls", bench._stage_calls(c))
    with torch.cuda.stream(st):
        for stage in order:
            calls[stage](st.cuda_stream)
        c["host_result"].copy_(c["result"], non_blocking=True)


def step_hilo(c, order, lo, hi):
    """cost, log-prob on `lo`; log-mel, DTW on `hi` (the DTW after its cost)."""
    calls = c.setdefault("_calls", bench._stage_calls(c))
    ev = c.setdefault("_ev_cost", torch.cuda.Event())
    ev2 = c.setdefault("_ev_dtw", torch.cuda.Event())
    for stage in order:
        if stage in ("cost", "logprob"):
            calls[stage](lo.cuda_stream)
            if stage == "cost":
                ev.record(lo)
        elif stage == "dtw":
            hi.wait_event(ev)
            calls[stage](hi.cuda_stream)
            ev2.record(hi)
        else:
            calls[stage](hi.cuda_stream)
    lo.wait_event(ev2)
    with torch.cuda.stream(lo):
        c["host_result"].copy_(c["result"], non_blocking=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--regions", type=int, default=25)
    ap.add_argument("--workload", default="kfull")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "overlap_matrix.json"))
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    w = bench.make_workload(dev, bench.WORKLOADS[args.workload], seed=1234)
    w["align"] = "split"
    rc, p_lo, p_hi = priority_range()
    results = {"priority_range": {"rc": rc, "least": p_lo, "greatest": p_hi}, "variants": []}
    print("priority range", rc, p_lo, p_hi, flush=True)
    sets = buffer_sets(w, dev, 4)
    # reference results: the plain schedule on buffer set 0
    step_serial(sets[0], ORDERS["same"][0], torch.cuda.current_stream())
    torch.cuda.synchronize()
    ref = sets[0]["host_result"].clone()

    variants = []
    for P in (1, 2, 3, 4):
        for order in ORDERS:
            if P == 1 and order != "same":
                continue
            for prio in ("none", "hilo", "hilo_noprio", "dtw_hi"):
                if prio != "none" and order not in ("same", "stagger_lp"):
                    continue
                variants.append((P, order, prio))
    if args.only:
        keep = set(args.only.split(","))
        variants = [v for v in variants if f"{v[0]}:{v[1]}:{v[2]}" in keep]

    for P, order, prio in variants:
        if prio == "none":
            streams = [(make_stream(dev),) for _ in range(P)]
        elif prio == "hilo":
            streams = [(make_stream(dev, p_lo), make_stream(dev, p_hi)) for _ in range(P)]
        elif prio == "hilo_noprio":
            streams = [(make_stream(dev), make_stream(dev)) for _ in range(P)]
        else:   # dtw_hi: serial order per batch, but every batch's stream is created high priority for odd batches
            streams = [(make_stream(dev, p_hi if j % 2 else p_lo),) for j in range(P)]

        def step(k):
            j = k % P
            c, o = sets[j], ORDERS[order][j]
            if len(streams[j]) == 1:
                step_serial(c, o, streams[j][0])
            else:
                step_hilo(c, o, streams[j][0], streams[j][1])
        for k in range(3 * P):
            step(k)
        torch.cuda.synchronize()
        ok = all(torch.equal(sets[j]["host_result"], ref) for j in range(P))
        regions = []
        for _ in range(args.regions):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(args.steps):
                step(k)
            torch.cuda.synchronize()
            regions.append((time.perf_counter() - t0) / args.steps * 1e3)
        rec = {"batches_in_flight": P, "order": order, "priorities": prio, "ms_per_step_median": round(float(np.median(regions)), 4),
               "ms_per_step_min": round(float(min(regions)), 4), "ms_per_step_max": round(float(max(regions)), 4),
               "results_identical": bool(ok)}
        print(json.dumps(rec), flush=True)
        results["variants"].append(rec)
        from whisper_timestamped import _lib
        for tup in streams:
            for s in tup:
                _lib.release_stream(s)
        del streams
    results["variants"].sort(key=lambda r: r["ms_per_step_median"])
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(results, open(args.out, "w"), indent=1)
    print("best:", json.dumps(results["variants"][:5], indent=1))


if __name__ == "__main__":
    main()
