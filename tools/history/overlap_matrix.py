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


def step_plan(c, plan, streams):
    """plan["assign"][stage] = (stream key, priority); the stages are ISSUED in plan["order"]; the DTW waits for its cost
    stage's event, the result copy for every stream of the step."""
    calls = c.setdefault("_calls", bench._stage_calls(c))
    ev = c.setdefault("_ev", {})
    used = []
    for stage in plan["order"]:
        st = streams[plan["assign"][stage][0]]
        if stage == "dtw":
            st.wait_event(ev["cost"])
        # the buffer set's previous step: its DTW has read the cost matrix this step's cost stage overwrites, its result copy
        # has read the record this step's DTW / log-prob gather overwrite (no-ops when both sit on the same stream)
        if stage == "cost" and "dtw" in ev:
            st.wait_event(ev["dtw"])
        if stage in ("dtw", "logprob") and "copy" in ev:
            st.wait_event(ev["copy"])
        calls[stage](st.cuda_stream)
        e = ev.get(stage)
        if e is None:
            e = ev[stage] = torch.cuda.Event()
        e.record(st)
        if st not in used:
            used.append(st)
    last = streams[plan["assign"][plan["order"][-1]][0]]
    for stage in plan["order"]:
        if streams[plan["assign"][stage][0]] is not last:
            last.wait_event(ev[stage])
    with torch.cuda.stream(last):
        c["host_result"].copy_(c["result"], non_blocking=True)
    e = ev.get("copy")
    if e is None:
        e = ev["copy"] = torch.cuda.Event()
    e.record(last)


H, N, L = "hi", "normal", "lo"
STD = ["logmel", "cost", "dtw", "logprob"]
PLANS = {
    "serial": dict(assign={s: ("a", N) for s in STD}, order=STD),
    "hilo": dict(assign={"logmel": ("h", H), "dtw": ("h", H), "cost": ("l", L), "logprob": ("l", L)}, order=STD),
    "hilo_lpfirst": dict(assign={"logmel": ("h", H), "dtw": ("h", H), "cost": ("l", L), "logprob": ("l", L)},
                         order=["logmel", "logprob", "cost", "dtw"]),
    "dtw_only_hi": dict(assign={"logmel": ("l", L), "dtw": ("h", H), "cost": ("l", L), "logprob": ("l", L)}, order=STD),
    "dtw_only_hi_lm_last": dict(assign={"logmel": ("l", L), "dtw": ("h", H), "cost": ("l", L), "logprob": ("l", L)},
                                order=["cost", "dtw", "logprob", "logmel"]),
    "hilo3": dict(assign={"logmel": ("n", N), "dtw": ("h", H), "cost": ("l", L), "logprob": ("l", L)}, order=STD),
    "hilo_lp_own": dict(assign={"logmel": ("h", H), "dtw": ("h", H), "cost": ("l", L), "logprob": ("l2", L)}, order=STD),
    "h2": dict(assign={"logmel": ("h1", H), "dtw": ("h2", H), "cost": ("l", L), "logprob": ("l", L)}, order=STD),
    "all4": dict(assign={"logmel": ("h1", H), "dtw": ("h2", H), "cost": ("l1", L), "logprob": ("l2", L)}, order=STD),
    "hi_vs_normal": dict(assign={"logmel": ("h", H), "dtw": ("h", H), "cost": ("l", N), "logprob": ("l", N)}, order=STD),
    "normal_vs_lo": dict(assign={"logmel": ("h", N), "dtw": ("h", N), "cost": ("l", L), "logprob": ("l", L)}, order=STD),
    "two_streams_no_prio": dict(assign={"logmel": ("h", N), "dtw": ("h", N), "cost": ("l", N), "logprob": ("l", N)}, order=STD),
    "lm_lo_dtw_hi_lp_own": dict(assign={"logmel": ("l2", L), "dtw": ("h", H), "cost": ("l", L), "logprob": ("l", L)}, order=STD),
}


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
    results = {"priority_range": {"rc": rc, "least": p_lo, "greatest": p_hi}, "variants": [],
               "plans": {k: {"assign": {s_: list(v) for s_, v in p["assign"].items()}, "issue_order": p["order"]} for k, p in PLANS.items()}}
    print("priority range", rc, p_lo, p_hi, flush=True)
    sets = buffer_sets(w, dev, 4)
    # reference results: the plain schedule on buffer set 0
    step_serial(sets[0], ORDERS["same"][0], torch.cuda.current_stream())
    torch.cuda.synchronize()
    ref = sets[0]["host_result"].clone()

    prio = {H: p_hi, N: 0, L: p_lo}
    variants = [(P, name) for P in (1, 2, 3) for name in PLANS]
    if args.only:
        keep = set(args.only.split(","))
        variants = [v for v in variants if f"{v[0]}:{v[1]}" in keep]
    from whisper_timestamped import _lib
    for P, name in variants:
        plan = PLANS[name]
        streams = []
        for j in range(P):
            d = {}
            for stage, (key, pr) in plan["assign"].items():
                if key not in d:
                    d[key] = make_stream(dev, prio[pr])
            streams.append(d)

        def step(k):
            j = k % P
            step_plan(sets[j], plan, streams[j])
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
        rec = {"batches_in_flight": P, "plan": name, "ms_per_step_median": round(float(np.median(regions)), 4),
               "ms_per_step_min": round(float(min(regions)), 4), "ms_per_step_max": round(float(max(regions)), 4),
               "ms_per_step_p90": round(float(np.percentile(regions, 90)), 4), "results_identical": bool(ok)}
        print(json.dumps(rec), flush=True)
        results["variants"].append(rec)
        for d in streams:
            for s_ in d.values():
                _lib.release_stream(s_)
        del streams
    results["variants"].sort(key=lambda r: r["ms_per_step_median"])
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(results, open(args.out, "w"), indent=1)
    print("best:", json.dumps(results["variants"][:5], indent=1))


if __name__ == "__main__":
    main()
