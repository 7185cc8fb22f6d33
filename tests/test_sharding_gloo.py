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
