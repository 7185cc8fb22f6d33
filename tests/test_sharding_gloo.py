"""N>1 path on the CPU: two processes, gloo backend (the same code runs on nccl = RCCL on the GPU node).

Units are sharded with no data-path collective; per-unit result records are gathered to rank 0, which assembles
words exactly as a single process would; weights are broadcast flat per dtype.  The kernels are replaced by the
CPU oracle here (tests/cpu_kernel_standin.py's numerics) -- what is under test is the sharding layer.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _units(n):
    import synth
    T, F = synth.draw_real_shapes(5, n)
    return [(int(t), int(f)) for t, f in zip(T, F)]


def _jumps_for(unit_index, T, F):
    """Deterministic stand-in for a unit's kernel output (oracle DTW on a seeded cost)."""
    from oracle import align_ref as O
    rng = np.random.RandomState(100 + unit_index)
    cost = -rng.rand(T, F)
    r = O.dtw_ref(cost)
    lp = -rng.rand(T).astype(np.float32)
    return O.jumps_from_path(r.index1s, r.index2s).astype(np.int32), lp


def _worker(rank, world, port, out_path):
    for p in (ROOT, os.path.join(ROOT, "whisper-timestamped_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from whisper_timestamped.sharding import ResultGatherer, broadcast_module_weights, partition_units
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # weights: rank 0 holds the truth, the others garbage
        torch.manual_seed(rank)
        net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.LayerNorm(5)).double()
        net.register_buffer("counter", torch.arange(3, dtype=torch.int64) + rank)
        broadcast_module_weights(dist, net, src=0)
        torch.manual_seed(0)
        ref = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.LayerNorm(5)).double()
        for a, b in zip(net.parameters(), ref.parameters()):
            assert torch.equal(a, b)
        assert torch.equal(net.counter, torch.arange(3, dtype=torch.int64))

        units = _units(23)
        parts = partition_units([t * f for t, f in units], world)
        assert sorted(i for p in parts for i in p) == list(range(len(units)))
        mine = parts[rank]
        # fixed-size records: every rank pads to the largest shard
        cap_j = max(sum(units[i][0] + 1 for i in p) for p in parts)
        cap_l = max(sum(units[i][0] for i in p) for p in parts)
        jumps = torch.full((cap_j,), -1, dtype=torch.int32)
        lps = torch.zeros(cap_l, dtype=torch.float32)
        oj = ol = 0
        for i in mine:
            T, F = units[i]
            j, lp = _jumps_for(i, T, F)
            jumps[oj:oj + T + 1] = torch.from_numpy(j)
            lps[ol:ol + T] = torch.from_numpy(lp)
            oj += T + 1
            ol += T
        g = ResultGatherer(dist, cap_j, cap_l, "cpu", every=2)
        for step in range(5):                     # double-buffered async path, 2 records per message (+ a partial one)
            g.gather(jumps if step == 4 else jumps + step + 1, lps)
        g.drain()                                 # the partial message (step 4) is record 0 of the newest gather
        if rank == 0:
            got = {}
            for r in range(world):
                bj, bl = g.unpack(r, step=0)
                oj = ol = 0
                for i in parts[r]:
                    T, F = units[i]
                    got[i] = (bj[oj:oj + T + 1].numpy().copy(), bl[ol:ol + T].numpy().copy())
                    oj += T + 1
                    ol += T
            for i, (T, F) in enumerate(units):
                j, lp = _jumps_for(i, T, F)
                assert np.array_equal(got[i][0], j) and np.array_equal(got[i][1], lp)
            open(out_path, "w").write("ok %d units over %d ranks" % (len(units), world))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_shard_and_gather(tmp_path):
    out = tmp_path / "result.txt"
    mp.spawn(_worker, args=(2, _free_port(), str(out)), nprocs=2, join=True)
    assert out.read_text().startswith("ok 23 units over 2 ranks")


@pytest.mark.timeout(600)
def test_eight_rank_shard_and_gather(tmp_path):
    """The node's shape (8 ranks, one per MI355X), on gloo: weights broadcast, 23 units dealt to 8 ranks, every rank's
    records gathered double-buffered and verified unit by unit on rank 0 (no 8-GPU node has been available
    to any round: this is what can be proven without one)."""
    out = tmp_path / "result.txt"
    mp.spawn(_worker, args=(8, _free_port(), str(out)), nprocs=8, join=True)
    assert out.read_text().startswith("ok 23 units over 8 ranks")


def test_partition_is_balanced_and_deterministic():
    sys.path.insert(0, os.path.join(ROOT, "whisper-timestamped_amd"))
    from whisper_timestamped.sharding import partition_units
    costs = [t * f for t, f in _units(200)]
    for world in (1, 2, 4, 8):
        parts = partition_units(costs, world)
        loads = [sum(costs[i] for i in p) for p in parts]
        assert sorted(i for p in parts for i in p) == list(range(200))
        assert max(loads) - min(loads) <= max(costs), (world, loads)      # LPT bound
        assert parts == partition_units(costs, world)


# ----------------------------------------------------------------------------------------------------------------------
# Long-form job sharded by speech island: every island must come out as the REFERENCE's own transcribe() of that crop
# (tests/golden/islands_job.json, written by tests/golden/make_golden_islands.py), for 1 and 2 ranks.
# ----------------------------------------------------------------------------------------------------------------------
def _islands_job():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "islands_job.json"), encoding="utf-8"))


def _run_islands_job(dist, job, patch, device="cpu", streams=0):
    import cpu_kernel_standin
    import whisper_double as W
    from golden import make_golden_transcribe as G
    from whisper_double.decoding import Script, set_row_scripts, set_script
    if patch is not None:
        cpu_kernel_standin.install(patch)
        if streams:
            from test_streams_host import install_streams_standin
            install_streams_standin(patch)
    W.install()
    from whisper_timestamped import streams as streams_mod
    from whisper_timestamped.sharding import transcribe_islands
    model, audio, _ = G.build_case(dict(job, script=None), device=device)
    rank = 0 if dist is None else dist.get_rank()
    if rank != 0:                      # rank 0 owns the truth: the others start from garbage weights and no audio
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
        audio = None
    seen = []

    def on_island(i):
        seen.append(i)
        set_script(Script(job["recorded"][i]))       # replay what the reference's decoder sampled on this island

    def on_batch(indices):                           # streams: one script per island, routed to its row of every decoder loop
        seen.extend(indices)
        scripts = [Script(job["recorded"][i]) for i in indices]

        def on_group(rows):                              # rows: positions in this rank's list of islands
            for r in rows:
                scripts[r].begin_window()
            set_row_scripts([scripts[r] for r in rows])
        streams_mod.ON_GROUP_DECODE = on_group
    try:
        result = transcribe_islands(model, audio, job["islands"], dist=dist, broadcast_weights=True, on_island=on_island,
                                    streams=streams, on_batch=on_batch, fp16=False, **job["opts"])
    finally:
        set_script(None)
        set_row_scripts(None)
        streams_mod.ON_GROUP_DECODE = None
    return result, seen


def _check_islands_result(result, job, time_tol=0.0, conf_tol=0.0):
    import json
    from golden import make_golden_transcribe as G
    from test_transcribe_host import compare
    from whisper_timestamped.sharding import merge_island_results
    islands = [tuple(x) for x in job["islands"]]
    got = json.loads(json.dumps(G.public_view(result), default=float))
    exp = G.public_view(merge_island_results(job["expected"], islands))
    exp = json.loads(json.dumps(exp, default=float))
    assert got["speech_activity"] == [{"start": s, "end": e} for s, e in islands]
    worst = compare(got, exp, time_tol=time_tol, conf_tol=conf_tol, logprob_tol=1e-6 if time_tol == 0 else 2e-4)
    # the merge itself, on numbers read off the golden: first word of island 1 sits 15.0 s after its crop-relative time
    n0 = len(job["expected"][0]["segments"])
    w_crop = job["expected"][1]["segments"][0]["words"][0]
    w_job = got["segments"][n0]["words"][0]
    assert w_job["text"] == w_crop["text"] and abs(w_job["start"] - (w_crop["start"] + 15.0)) <= time_tol + 1e-9
    assert got["segments"][n0]["seek"] == job["expected"][1]["segments"][0]["seek"] + 1500
    assert [s["id"] for s in got["segments"]] == list(range(len(got["segments"])))
    starts = [s["start"] for s in got["segments"]]
    assert starts == sorted(starts)
    return worst


def test_islands_job_single_rank(monkeypatch):
    job = _islands_job()
    result, seen = _run_islands_job(None, job, monkeypatch)
    assert seen == [0, 1, 2, 3]
    _check_islands_result(result, job)


def test_islands_job_single_rank_islands_as_decoder_streams(monkeypatch):
    """The same job with the rank's islands stepping through the decoder together (streams = 4): per island the
    reference's output, as one after the other."""
    job = _islands_job()
    result, seen = _run_islands_job(None, job, monkeypatch, streams=4)
    assert sorted(seen) == [0, 1, 2, 3]
    _check_islands_result(result, job)


def _islands_worker(rank, world, port, out_path, streams=0):
    for p in (ROOT, os.path.join(ROOT, "whisper-timestamped_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(4)                        # (two ranks share the host's cores)
    patch = pytest.MonkeyPatch()
    try:
        job = _islands_job()
        result, seen = _run_islands_job(dist, job, patch, streams=streams)
        owned = [None] * world
        dist.all_gather_object(owned, seen)
        assert sorted(i for part in owned for i in part) == [0, 1, 2, 3] and all(len(part) > 0 for part in owned)
        if rank == 0:
            _check_islands_result(result, job)
            open(out_path, "w").write("ok " + repr(owned))
        else:
            assert result is None
        dist.barrier()
    finally:
        patch.undo()
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_islands_job_two_ranks(tmp_path):
    out = tmp_path / "islands.txt"
    mp.spawn(_islands_worker, args=(2, _free_port(), str(out)), nprocs=2, join=True)
    assert out.read_text().startswith("ok ")


@pytest.mark.timeout(600)
def test_islands_job_two_ranks_islands_as_decoder_streams(tmp_path):
    """BASELINE configs[3] in its MI355X shape: islands dealt to the ranks (no data-path collective), and each rank's
    islands decoded together, several streams per decoder op."""
    out = tmp_path / "islands_streams.txt"
    mp.spawn(_islands_worker, args=(2, _free_port(), str(out), 4), nprocs=2, join=True)
    assert out.read_text().startswith("ok ")


def _recordings_worker(rank, world, port, out_path):
    """sharding.transcribe_recordings on two gloo ranks: the four same-model goldens dealt largest-first, each rank decodes
    its share as decoder streams, rank 0 gets the four dictionaries in order -- each equal to the reference's output for
    that recording."""
    for p in (ROOT, os.path.join(ROOT, "whisper-timestamped_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import json
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    patch = pytest.MonkeyPatch()
    try:
        import cpu_kernel_standin
        import whisper_double as W
        from golden import make_golden_transcribe as G
        from test_streams_host import install_streams_standin, same_model_cases
        from test_transcribe_host import compare, rounded
        from whisper_double.decoding import Script, set_row_scripts
        cpu_kernel_standin.install(patch)
        install_streams_standin(patch)
        W.install()
        from whisper_timestamped import streams, words
        from whisper_timestamped.sharding import transcribe_recordings
        torch.set_num_threads(4)                    # (two ranks share the host's cores)
        cases = same_model_cases()
        model, _, _ = G.build_case(cases[0], device="cpu")
        if rank != 0:
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(1.0)                     # garbage until rank 0's weights arrive
        audios = [G.build_case(c, device="cpu")[1] for c in cases]
        seen = []

        def on_batch(indices):
            seen.extend(indices)
            scripts = [Script(cases[i]["recorded"]) for i in indices]

            def on_group(rows):
                for r in rows:
                    scripts[r].begin_window()
                set_row_scripts([scripts[r] for r in rows])
            streams.ON_GROUP_DECODE = on_group
        words.RAW_CONFIDENCE = True
        try:
            results = transcribe_recordings(model, audios, dist=dist, broadcast_weights=True, streams=8, on_batch=on_batch,
                                            fp16=False, **cases[0]["opts"])
        finally:
            words.RAW_CONFIDENCE = False
            streams.ON_GROUP_DECODE = None
            set_row_scripts(None)
        owned = [None] * world
        dist.all_gather_object(owned, seen)
        assert sorted(i for part in owned for i in part) == list(range(4)) and all(len(part) > 0 for part in owned)
        if rank == 0:
            for r, c in zip(results, cases):
                view = json.loads(json.dumps(G.public_view(r), default=float))
                compare(rounded(view), c["expected"], time_tol=0.0, conf_tol=0.0, logprob_tol=1e-5)
            open(out_path, "w").write("ok " + repr(owned))
        else:
            assert results is None
        dist.barrier()
    finally:
        patch.undo()
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_recordings_across_two_ranks_streams_within_a_rank(tmp_path):
    out = tmp_path / "recordings.txt"
    mp.spawn(_recordings_worker, args=(2, _free_port(), str(out)), nprocs=2, join=True)
    assert out.read_text().startswith("ok ")


def _collect_worker(rank, world, port, out_path):
    """sharding.collect_results: every rank's [(index, dictionary)] to rank 0 as flat records in one fixed-size tensor gather
    ("dicts", "packed") or as pickles ("pickle", rounds 1-5): the same dictionaries in the caller's order, ranks with
    different numbers of results (one of them with none)."""
    for p in (ROOT, os.path.join(ROOT, "whisper-timestamped_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import json
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from whisper_timestamped.sharding import collect_results
        cases = json.load(open(os.path.join(ROOT, "tests", "golden", "transcribe_cases.json")))
        pool = [c["expected"] for c in cases][:11]
        owner = [0, 1, 1, 0, 1, 1, 0, 1, 1, 1, 0] if world == 2 else [0, 2, 2, 0, 2, 2, 0, 2, 2, 2, 0]     # (3 ranks: rank 1 owns nothing)
        mine = [(i, pool[i]) for i in range(len(pool)) if owner[i] == rank]
        for mode in ("dicts", "packed", "pickle"):
            got = collect_results(dist, mine, len(pool), "cpu", mode)
            if rank != 0:
                assert got is None
                continue
            if mode == "packed":
                assert len(got) == len(pool) and got.dict(4) == pool[4]
                got = got.dicts()
            assert got == pool, mode
        if rank == 0:
            open(out_path, "w").write("ok")
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_results_travel_as_flat_records_or_pickles_and_arrive_the_same(tmp_path, world):
    out = tmp_path / "collect.txt"
    mp.spawn(_collect_worker, args=(world, _free_port(), str(out)), nprocs=world, join=True)
    assert out.read_text() == "ok"


def test_transcribe_many_two_worker_processes_on_the_cpu(monkeypatch):
    """sharding.transcribe_many's process plumbing (spawned workers, largest-first dealing, common start, results back in
    the caller's order) with the kernels replaced by the oracle-backed stand-in: the dictionaries of serial transcribe()
    calls in this process.  The same comparison with the real kernels: tests/test_gpu_transcribe.py."""
    import torch
    import many_helper as H
    import cpu_kernel_standin
    import whisper_timestamped as wt
    from whisper_double.decoding import set_script
    from whisper_timestamped.sharding import transcribe_many
    g = torch.Generator().manual_seed(11)
    audios = [(0.05 * torch.randn(n, generator=g)).float() for n in (30 * 16000, 17 * 16000)]
    cpu_kernel_standin.install(monkeypatch)
    model = H.load_tiny("cpu")
    serial = []
    for k, a in enumerate(audios):
        H.script_clip(k)
        serial.append(wt.transcribe(model, a, language="en", fp16=False))
    set_script(None)
    many = transcribe_many(H.load_tiny, audios, workers_per_gpu=2, devices=["cpu"], on_item=H.script_clip_cpu, language="en",
                           fp16=False)
    assert len(many) == len(serial)
    for a, b in zip(many, serial):
        assert a["text"] == b["text"] and len(a["segments"]) == len(b["segments"]) > 0
        for sa, sb in zip(a["segments"], b["segments"]):
            assert sa["words"] == sb["words"]


@pytest.mark.timeout(600)
def test_transcribe_many_worker_processes_with_decoder_streams_and_warm_up(monkeypatch):
    """transcribe_many(streams=2, warmup=True): every worker warms up on ONE recording of its batch through the
    B-stream driver (not on its whole batch), meets the others at the barrier (timeout a parameter), then decodes its
    recordings together -- same dictionaries as serial transcribe() calls."""
    import torch
    import many_helper as H
    import cpu_kernel_standin
    import whisper_timestamped as wt
    from whisper_double.decoding import set_script
    from whisper_timestamped.sharding import transcribe_many
    g = torch.Generator().manual_seed(12)
    audios = [(0.05 * torch.randn(n, generator=g)).float() for n in (30 * 16000, 21 * 16000, 26 * 16000, 18 * 16000)]
    cpu_kernel_standin.install(monkeypatch)
    model = H.load_tiny("cpu")
    serial = []
    for k, a in enumerate(audios):
        H.script_clip(k)
        serial.append(wt.transcribe(model, a, language="en", fp16=False))
    set_script(None)
    many, seconds = transcribe_many(H.load_tiny, audios, workers_per_gpu=2, devices=["cpu"], streams=2, warmup=True,
                                    on_batch=H.script_batch_cpu, return_timing=True, barrier_timeout=300.0, language="en", fp16=False)
    assert len(many) == len(serial) and seconds > 0
    for a, b in zip(many, serial):
        assert a["text"] == b["text"] and len(a["segments"]) == len(b["segments"]) > 0
        for sa, sb in zip(a["segments"], b["segments"]):
            assert [(w["text"], w["start"], w["end"]) for w in sa["words"]] == [(w["text"], w["start"], w["end"]) for w in sb["words"]]
