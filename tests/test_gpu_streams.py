"""Several batches in flight on distinct HIP streams (the configuration bench.py defaults to): every stream owns its
inputs, outputs and -- inside the library -- its scratch arenas.  Each buffer set is first run alone (its reference;
set 0's reference is held against the oracle), then 2 and 3 sets run concurrently for >= 200 steps and every set must
reproduce its own reference bit for bit at every checkpoint: a scratch arena shared between streams, a DTW plane slot
written by the wrong workgroup or a stale result record shows up as a difference."""
import numpy as np
import pytest
import torch

from oracle import align_ref as O

import workloads as WL

pytestmark = pytest.mark.gpu


def _set(dev, seed, n_chunks):
    cfg = dict(WL.WORKLOADS["kfull"], n_chunks=n_chunks)
    return WL.make_workload(dev, cfg, seed=seed)


def _snapshot(w):
    # (the cost buffer ends with the 16 bytes of read slack include/wtalign.h asks for: never written, not compared)
    n_cost = max(int(d["cost_offset"]) + int(d["T"]) * int(d["F"]) for d in w["descs"])
    snap = {k: w[k].clone() for k in ("result", "mel", "pad", "gmax")}
    snap["cost"] = w["cost"][:n_cost].clone()
    return snap


@pytest.mark.parametrize("n_sets,n_chunks,steps", [(2, 32, 200), (3, 32, 210), (3, 5, 300)])
def test_batches_in_flight_on_distinct_streams(n_sets, n_chunks, steps):
    dev = torch.device("cuda", 0)
    sets = [_set(dev, 900 + j, n_chunks) for j in range(n_sets)]
    refs = []
    for w in sets:                                   # each set alone, on the default stream
        WL.run_step(w)
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
            WL.run_step(sets[j])
        if k % 50 == 49 or k == steps - 1:           # checkpoint: every set equals its own single-stream reference
            torch.cuda.synchronize()
            for j2, (w, ref) in enumerate(zip(sets, refs)):
                if k < n_sets and j2 > k:
                    continue
                for name, want in ref.items():
                    assert torch.equal(w[name][:want.shape[0]], want), f"step {k}: buffer set {j2}: {name} differs from its single-stream reference"
                assert torch.equal(w["host_result"], ref["result"].cpu())


@pytest.mark.parametrize("schedule,n_sets,order", [("hilo", 2, "round_robin"), ("hilo", 3, "round_robin"), ("serial", 2, "round_robin"),
                                                   ("hilo", 2, "pairs"), ("hilo", 2, "random")])
def test_pipeline_schedules_reproduce_the_single_stream_results(schedule, n_sets, order):
    """whisper_timestamped.pipeline.HotPathPipeline (what bench.py's timed region and batched.py run): the stages of a batch
    on a high- and a low-priority HIP stream, the DTW behind its cost stage by event, a batch's next step behind what still
    reads its buffers.  Buffer sets with their OWN inputs, >= 120 steps: every set equals its single-stream reference bit
    for bit at every checkpoint -- a missing cross-stream dependency (the DTW reading a cost matrix that is being
    rewritten, a result record copied before the DTW has written it) shows up as a difference.  `pairs` / `random` submit
    the batches in an order that does NOT keep a batch on one stream set: the dependencies travel with the buffers."""
    from whisper_timestamped.pipeline import HotPathPipeline
    dev = torch.device("cuda", 0)
    sets = [_set(dev, 700 + j, 32) for j in range(n_sets)]
    refs = []
    for w in sets:
        WL.run_step(w)
        torch.cuda.synchronize()
        refs.append(_snapshot(w))
    pipe = HotPathPipeline(dev, depth=n_sets, schedule=schedule)
    assert pipe.describe()["schedule"] == schedule
    for w in sets:
        for k in ("result", "mel", "cost"):
            w[k].zero_()
        w["host_result"].zero_()
    torch.cuda.synchronize()
    steps = 120
    rs = np.random.RandomState(5)
    for k in range(steps):
        j = {"round_robin": k % n_sets, "pairs": (k // 2) % n_sets, "random": int(rs.randint(n_sets))}[order]
        pipe.submit(sets[j]["batch"])
        if k % 40 == 39 or k == steps - 1:
            torch.cuda.synchronize()
            for j2, (w, ref) in enumerate(zip(sets, refs)):
                for name, want in ref.items():
                    assert torch.equal(w[name][:want.shape[0]], want), f"{schedule}, step {k}: buffer set {j2}: {name} differs"
                assert torch.equal(w["host_result"], ref["result"].cpu()), f"{schedule}, step {k}: buffer set {j2}: host record differs"
    pipe.release()


@pytest.mark.parametrize("n_chunks,parts,depth", [(32, 4, 2), (48, 3, 2), (40, 4, 3)])
def test_pipeline_sub_batches_equal_the_whole_batch(n_chunks, parts, depth):
    """HotPathPipeline(sub_batches=P): a batch launched as P chunk ranges round-robin over the stream sets, ONE result copy
    behind all of them -- bit-identical to the batch launched whole on one stream, two buffer sets over the same inputs in
    flight, 60 steps."""
    from whisper_timestamped.pipeline import HotPathPipeline
    dev = torch.device("cuda", 0)
    w = _set(dev, 811, n_chunks)
    WL.run_step(w)
    torch.cuda.synchronize()
    ref = _snapshot(w)
    sets = [w, WL.twin(w)]
    pipe = HotPathPipeline(dev, depth=depth, schedule="hilo", sub_batches=parts, rows_per_chunk=w["cfg"]["T"])
    for c in sets:
        for k in ("result", "mel", "cost"):
            c[k].zero_()
        c["host_result"].zero_()
    torch.cuda.synchronize()
    for k in range(60):
        pipe.submit(sets[k % 2]["batch"])
        if k % 20 == 19:
            torch.cuda.synchronize()
            for c in sets:
                for name, want in ref.items():
                    assert torch.equal(c[name][:want.shape[0]], want), f"step {k}: {name} differs from the whole-batch launch"
                assert torch.equal(c["host_result"], ref["result"].cpu())
    pipe.release()
