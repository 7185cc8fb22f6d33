"""Several batches in flight on distinct HIP streams (the configuration bench.py defaults to): every stream owns its
inputs, outputs and -- inside the library -- its scratch arenas.  Each buffer set is first run alone (its reference;
set 0's reference is held against the oracle), then 2 and 3 sets run concurrently for >= 200 steps and every set must
reproduce its own reference bit for bit at every checkpoint: a scratch arena shared between streams, a DTW plane slot
written by the wrong workgroup or a stale result record shows up as a difference."""
import numpy as np
import pytest
import torch

from oracle import align_ref as O

pytestmark = pytest.mark.gpu


def _set(bench, dev, seed, n_chunks):
    cfg = dict(bench.WORKLOADS["kfull"], n_chunks=n_chunks)
    return bench.make_workload(dev, cfg, seed=seed)


def _snapshot(w):
    # (the cost buffer ends with the 16 bytes of read slack include/wtalign.h asks for: never written, not compared)
    n_cost = max(int(d["cost_offset"]) + int(d["T"]) * int(d["F"]) for d in w["descs"])
    snap = {k: w[k].clone() for k in ("result", "mel", "pad", "gmax")}
    snap["cost"] = w["cost"][:n_cost].clone()
    return snap


@pytest.mark.parametrize("n_sets,n_chunks,steps", [(2, 32, 200), (3, 32, 210), (3, 5, 300)])
def test_batches_in_flight_on_distinct_streams(n_sets, n_chunks, steps):
    import bench
    dev = torch.device("cuda", 0)
    sets = [_set(bench, dev, 900 + j, n_chunks) for j in range(n_sets)]
    refs = []
    for w in sets:                                   # each set alone, on the default stream
        bench.run_step(w)
        torch.cuda.synchronize()
        refs.append(_snapshot(w))
    # set 0 against the oracle: DTW jumps bit-exact for the GPU's cost matrix, cost and log-probs within the parity bars
    w, cfg = sets[0], sets[0]["cfg"]
    T, F = cfg["T"], cfg["F"]
    jumps = w["jumps"].cpu().numpy()
    for b in (0, n_chunks - 1):
        d = w["descs"][b]
        c = w["cost"][int(d["cost_offset"]):int(d["cost_offset"]) + T * F].reshape(T, F).cpu().numpy()
        ref = O.cost_matrix_ref(w["qk"][b].float().cpu(), 9, 1.0, None, 0)
        assert np.abs(c - ref).max() / np.abs(ref).max() < 2e-6
        r = O.dtw_ref(c.astype(np.float64))
        assert np.array_equal(jumps[int(d["jumps_offset"]):int(d["jumps_offset"]) + T + 1], O.jumps_from_path(r.index1s, r.index2s))
        lp = w["logprob"][b * T:(b + 1) * T].cpu()
        want = O.token_logprob_gather_ref(w["logits"][b * T:(b + 1) * T].cpu(), w["tokens"][b * T:(b + 1) * T].cpu().numpy())
        assert (lp - want).abs().max() < 2e-5
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_sets)]
    for w in sets:
        for k in ("result", "mel", "cost"):
            w[k].zero_()
    torch.cuda.synchronize()
    for k in range(steps):
        j = k % n_sets
        with torch.cuda.stream(streams[j]):
            bench.run_step(sets[j])
        if k % 50 == 49 or k == steps - 1:           # checkpoint: every set equals its own single-stream reference
            torch.cuda.synchronize()
            for j2, (w, ref) in enumerate(zip(sets, refs)):
                if k < n_sets and j2 > k:
                    continue
                for name, want in ref.items():
                    assert torch.equal(w[name][:want.shape[0]], want), f"step {k}: buffer set {j2}: {name} differs from its single-stream reference"
                assert torch.equal(w["host_result"], ref["result"].cpu())


@pytest.mark.parametrize("schedule,n_sets", [("hilo", 2), ("hilo", 3), ("two_streams", 2), ("dtw_hi", 2), ("hilo_one_lo", 2)])
def test_stream_priority_schedules_reproduce_the_single_stream_results(schedule, n_sets):
    """bench.py --schedule (round 5): the stages of a batch on a high- and a low-priority HIP stream, the DTW behind its
    cost stage by event, a buffer set's next step behind what still reads its buffers.  Buffer sets with their OWN inputs,
    >= 120 steps: every set equals its single-stream reference bit for bit at every checkpoint -- a missing cross-stream
    dependency (the DTW reading a cost matrix that is being rewritten, a result record copied before the DTW has written
    it) shows up as a difference."""
    import bench
    dev = torch.device("cuda", 0)
    sets = [_set(bench, dev, 700 + j, 32) for j in range(n_sets)]
    refs = []
    for w in sets:
        bench.run_step(w)
        torch.cuda.synchronize()
        refs.append(_snapshot(w))
    plan = bench.SCHEDULES[schedule]
    shared = {}
    stream_sets = [bench.plan_streams(dev, plan, shared) for _ in range(n_sets)]
    for w in sets:
        for k in ("result", "mel", "cost"):
            w[k].zero_()
        w["host_result"].zero_()
    torch.cuda.synchronize()
    steps = 120
    for k in range(steps):
        j = k % n_sets
        bench.run_step_plan(sets[j], plan, stream_sets[j])
        if k % 40 == 39 or k == steps - 1:
            torch.cuda.synchronize()
            for j2, (w, ref) in enumerate(zip(sets, refs)):
                for name, want in ref.items():
                    assert torch.equal(w[name][:want.shape[0]], want), f"{schedule}, step {k}: buffer set {j2}: {name} differs"
                assert torch.equal(w["host_result"], ref["result"].cpu()), f"{schedule}, step {k}: buffer set {j2}: host record differs"
    from whisper_timestamped import _lib
    for d in stream_sets:
        for s in d.values():
            _lib.release_stream(s)
