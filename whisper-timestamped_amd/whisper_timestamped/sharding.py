"""Multi-GPU sharding of alignment units (one process per GPU, RCCL over xGMI).

The reference is single-process (no communication layer).
Alignment units are independent, so the MI355X design is: every rank owns a
subset of the units (largest-first balancing on T*F, since one DTW's latency
grows with T+F), runs the kernels with NO data-path collective, and the
KB-sized per-unit records (jumps[T+1] int32, log-probs[T] fp32) are gathered
to rank 0, which assembles words/JSON.  Weights are replicated (broadcast once).
Works with any torch.distributed backend: nccl (= RCCL) on GPUs, gloo in the
CPU tests.
"""
from __future__ import annotations

import heapq

import torch


def partition_units(costs, world_size: int):
    """Longest-processing-time-first assignment.  costs[i] ~ T_i*F_i.
    Returns a list of index lists, one per rank (deterministic)."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    heap = [(0, r) for r in range(world_size)]
    heapq.heapify(heap)
    parts = [[] for _ in range(world_size)]
    for i in order:
        load, r = heapq.heappop(heap)
        parts[r].append(i)
        heapq.heappush(heap, (load + costs[i], r))
    for p in parts:
        p.sort()
    return parts


def broadcast_module_weights(dist, module: torch.nn.Module, src: int = 0):
    """One flat broadcast per dtype (few large messages: xGMI links are
    point-to-point, per-message latency dominates small sends)."""
    by_dtype = {}
    for t in list(module.parameters()) + list(module.buffers()):
        if t.is_sparse:
            continue
        by_dtype.setdefault(t.dtype, []).append(t)
    for dt, tensors in by_dtype.items():
        flat = torch.cat([t.detach().reshape(-1) for t in tensors])
        dist.broadcast(flat, src=src)
        off = 0
        for t in tensors:
            n = t.numel()
            t.data.copy_(flat[off:off + n].view_as(t))
            off += n


class ResultGatherer:
    """Fixed-size gather of per-rank result records to rank 0 (preallocated
    buffers: nothing is allocated per step)."""

    def __init__(self, dist, n_jumps: int, n_logprob: int, device):
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.n_jumps, self.n_logprob = n_jumps, n_logprob
        # one int32 record: jumps followed by the log-probs' bit patterns
        self.send = torch.empty(n_jumps + n_logprob, dtype=torch.int32, device=device)
        self.recv = ([torch.empty_like(self.send) for _ in range(self.world)] if self.rank == 0 else None)

    def gather(self, jumps: torch.Tensor, logprob: torch.Tensor):
        self.send[: self.n_jumps].copy_(jumps)
        self.send[self.n_jumps:].copy_(logprob.view(torch.int32))
        self.dist.gather(self.send, self.recv, dst=0)
        return self.recv

    def unpack(self, r: int):
        buf = self.recv[r]
        return buf[: self.n_jumps], buf[self.n_jumps:].view(torch.float32)
