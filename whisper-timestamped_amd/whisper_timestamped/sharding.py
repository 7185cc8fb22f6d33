"""Multi-GPU sharding of alignment units (one process per GPU, RCCL over xGMI).

The reference is single-process (no communication layer).
Alignment units are independent, so the MI355X design is: every rank owns a
subset of the units (largest-first balancing on T*F, since one DTW's latency
grows with T+F), runs the kernels with NO data-path collective, and the
KB-sized per-unit records (jumps[T+1] int32, log-probs[T] fp32) are gathered
to rank 0, which assembles words/JSON; finished transcribe() dictionaries travel
as one byte record per recording in one fixed-size tensor gather (records.py),
decoded on rank 0 when they are read.  Weights are replicated (broadcast once).
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
    """Fixed-size gather of per-rank result records to rank 0.

    Few, larger collectives: the records of ``every`` consecutive steps are packed into one message (xGMI links are
    point-to-point and a tiny gather is pure latency: ~60 us per call measured, against a 0.64 ms step).  Buffers are
    preallocated and DOUBLE-BUFFERED: ``gather`` launches the collective asynchronously when a message is full; it
    overlaps the kernels of the following steps, and a buffer is only rewritten after the collective that read it
    has been waited for (stream-side wait on the GPU, no host sync).  ``flush`` sends a partial message;
    ``latest()`` waits for the newest collective and returns its receive list (rank 0)."""

    def __init__(self, dist, n_jumps: int, n_logprob: int, device, depth: int = 2, every: int = 1):
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.n_jumps, self.n_logprob, self.every = n_jumps, n_logprob, every
        self.rec = n_jumps + n_logprob          # one int32 record: jumps followed by the log-probs' bit patterns
        self.send = [torch.zeros(every * self.rec, dtype=torch.int32, device=device) for _ in range(depth)]
        self.recv = [([torch.empty_like(self.send[0]) for _ in range(self.world)] if self.rank == 0 else None)
                     for _ in range(depth)]
        self.work = [None] * depth
        self.k = 0            # buffer being filled
        self.fill = 0         # records already in it
        self.sent = -1        # buffer of the newest collective

    def gather(self, jumps: torch.Tensor, logprob: torch.Tensor):
        k = self.k
        if self.fill == 0 and self.work[k] is not None:
            self.work[k].wait()              # the collective that last read this buffer
            self.work[k] = None
        rec = self.send[k][self.fill * self.rec:(self.fill + 1) * self.rec]
        rec[: self.n_jumps].copy_(jumps)
        rec[self.n_jumps:].copy_(logprob.view(torch.int32))
        self.fill += 1
        if self.fill == self.every:
            return self.flush()
        return None

    def flush(self):
        if self.fill == 0:
            return None
        k = self.k
        self.work[k] = self.dist.gather(self.send[k], self.recv[k], dst=0, async_op=True)
        self.sent = k
        self.k = (k + 1) % len(self.send)
        self.fill = 0
        return self.work[k]

    def latest(self):
        if self.sent >= 0 and self.work[self.sent] is not None:
            self.work[self.sent].wait()
        return self.recv[self.sent]

    def drain(self):
        self.flush()
        for w in self.work:
            if w is not None:
                w.wait()

    def unpack(self, r: int, step: int = -1):
        """(jumps, logprob) of rank r for record `step` of the newest message (-1 = its last record)."""
        buf = self.latest()[r]
        if step < 0:
            step += self.every
        rec = buf[step * self.rec:(step + 1) * self.rec]
        return rec[: self.n_jumps], rec[self.n_jumps:].view(torch.float32)


# ----------------------------------------------------------------------------------------------------------------------
# Long-form jobs: VAD islands as the sharding unit (BASELINE config 4)
# ----------------------------------------------------------------------------------------------------------------------
def _shift(x, t0):
    return round(x + t0, 2)


def merge_island_results(results, islands):
    """results[i]: the transcribe() dictionary of island i, on the time axis of ITS crop; islands[i] = (start_s, end_s)
    in the full recording.  Returns one dictionary on the time axis of the recording: texts concatenated, segment ids
    renumbered, `seek` (10 ms mel frames) / segment / word times shifted by the island start, `speech_activity` = the
    islands (the key the reference fills when VAD is on, transcribe.py:352-355).  `language` / `language_probs` are
    those of the first island."""
    out = {"text": "", "segments": [], "language": None}
    for (t0, _t1), r in zip(islands, results):
        if out["language"] is None:
            out["language"] = r.get("language")
            if "language_probs" in r:
                out["language_probs"] = r["language_probs"]
        out["text"] += r["text"]
        for seg in r["segments"]:
            seg = dict(seg)
            seg["id"] = len(out["segments"])
            seg["seek"] = seg["seek"] + int(round(t0 * 100))
            seg["start"], seg["end"] = _shift(seg["start"], t0), _shift(seg["end"], t0)
            if "words" in seg:
                seg["words"] = [dict(w, start=_shift(w["start"], t0), end=_shift(w["end"], t0)) for w in seg["words"]]
            out["segments"].append(seg)
    out["speech_activity"] = [{"start": s, "end": e} for (s, e) in islands]
    return out


def share_audio(dist, audio, device, src: int = 0):
    """Rank `src` holds the recording (1-D fp32); the others may pass None and receive it (one broadcast)."""
    if dist is None or dist.get_world_size() == 1:
        return audio
    n = torch.tensor([0 if audio is None else audio.numel()], dtype=torch.int64, device=device)
    dist.broadcast(n, src=src)
    buf = audio.to(device=device, dtype=torch.float32) if dist.get_rank() == src \
        else torch.empty(int(n.item()), dtype=torch.float32, device=device)
    dist.broadcast(buf, src=src)
    return buf


def transcribe_islands(model, audio, islands, dist=None, broadcast_weights: bool = False, on_island=None,
                       sample_rate: int = 16000, streams: int = 0, on_batch=None, **options):
    """One long recording, many ranks.  `islands` = [(start_s, end_s)] speech islands (an explicit VAD list: the
    reference's `vad=[...]` form, transcribe.py:1944-1947); every island is an independent unit -- exactly the
    reference's transcribe() on that island's crop -- so they are dealt to the ranks largest-first by duration with
    NO data-path collective; every rank transcribes its islands on its own GPU and rank 0 receives the per-island
    dictionaries (byte records in one fixed-size tensor gather per job, records.py) and merges them (merge_island_results).  Non-zero ranks return None.

    Not the reference's `vad=` mode (that one glues the islands and decodes them as ONE stream, which cannot be split
    without changing what the decoder is conditioned on); per island the results are the reference's.
    `audio`: what transcribe() accepts (path, ndarray, 1-D tensor), on rank 0 (None elsewhere -> broadcast) or on every rank.  `on_island(i)` is called
    before island i is transcribed (progress / test scripting).  `options` go to transcribe_timestamped().
    `streams` > 1: a rank's islands are independent recordings, so they step through the decoder TOGETHER, up to
    `streams` of them per decoder op (transcribe_batch -> streams.py) instead of one after the other; `on_batch(indices)`
    is then called once per rank with its island indices in stream order.  Same per-island results."""
    from .naive import get_audio_tensor
    from .transcribe import transcribe_timestamped
    if audio is not None:
        audio = get_audio_tensor(audio)            # a path, an ndarray or a tensor, like transcribe()
    rank = 0 if dist is None else dist.get_rank()
    world = 1 if dist is None else dist.get_world_size()
    islands = [(float(s), float(e)) for s, e in islands]
    assert all(e > s >= 0 for s, e in islands), "islands must be (start, end) pairs in seconds with end > start"
    if world > 1:
        if broadcast_weights:
            broadcast_module_weights(dist, model, src=0)
        audio = share_audio(dist, audio, model.device)
    parts = partition_units([e - s for s, e in islands], world)
    mine = []
    if streams and streams > 1:
        from .transcribe import transcribe_batch
        idx = list(parts[rank])
        crops = [audio[int(round(islands[i][0] * sample_rate)):int(round(islands[i][1] * sample_rate))] for i in idx]
        if on_batch is not None:
            on_batch(idx)
        mine = list(zip(idx, transcribe_batch(model, crops, max_streams=streams, **options)))
    else:
        for i in parts[rank]:
            s, e = islands[i]
            crop = audio[int(round(s * sample_rate)):int(round(e * sample_rate))]
            if on_island is not None:
                on_island(i)
            mine.append((i, transcribe_timestamped(model, crop, **options)))
    per_island = collect_results(dist, mine, len(islands), model.device, "dicts")       # (byte records, one tensor gather)
    if per_island is None:
        return None
    return merge_island_results(per_island, islands)


# ----------------------------------------------------------------------------------------------------------------------
# Many recordings, many ranks: recordings across the GPUs, decoder streams within a GPU
# ----------------------------------------------------------------------------------------------------------------------
def transcribe_recordings(model, audios, dist=None, broadcast_weights: bool = False, streams: int = 32, on_batch=None,
                          results: str = "dicts", **options):
    """transcribe_timestamped() of every recording in `audios` on all ranks of `dist` (one process per GPU).  Recordings
    are independent units: dealt to the ranks largest-first by length (`partition_units`), no data-path collective;
    each rank steps ITS recordings through the decoder together, up to `streams` per decoder op (`transcribe_batch`);
    rank 0 receives the results and returns them in the order of `audios` (the other ranks return None).  Every rank
    passes the same `audios` list (paths, arrays or tensors; a rank only loads its own).  `on_batch(indices)` is called on
    each rank with its recordings' indices, in stream order.

    How the results travel (`results`; measured at 8 ranks x 32 recordings, tools/measure_result_gather.py):
      "dicts"   one byte record per recording in ONE fixed-size tensor gather (records.gather_packed), every record decoded
                on rank 0 -- the same list of dictionaries transcribe() would give, one per recording;
      "packed"  the same gather, and rank 0 returns the records.PackedResults table: a dictionary is built when somebody
                asks for it (``table.dict(i)``), so rank 0's serial share of the job is the gather alone;
      "pickle"  rounds 1-5: one ``dist.gather_object`` per job (every object of every rank rebuilt inside the collective)."""
    assert results in ("dicts", "packed", "pickle"), results
    from .naive import get_audio_tensor
    from .transcribe import transcribe_batch
    rank = 0 if dist is None else dist.get_rank()
    world = 1 if dist is None else dist.get_world_size()
    audios = list(audios)
    if world > 1 and broadcast_weights:
        broadcast_module_weights(dist, model, src=0)

    def length(a):
        if isinstance(a, str):
            import os
            return os.path.getsize(a)            # (a proxy: compressed files of one codec scale with their duration)
        return int(a.shape[-1])
    parts = partition_units([length(a) for a in audios], world)
    idx = list(parts[rank])
    if on_batch is not None:
        on_batch(idx)
    mine = list(zip(idx, transcribe_batch(model, [get_audio_tensor(audios[i]) for i in idx], max_streams=streams, **options)))
    return collect_results(dist, mine, len(audios), model.device, results)


def collect_results(dist, mine, n_total, device, results="dicts"):
    """[(index, transcribe() dictionary)] of every rank -> rank 0 (see transcribe_recordings); None on the other ranks."""
    rank = 0 if dist is None else dist.get_rank()
    world = 1 if dist is None else dist.get_world_size()
    if world > 1 and results == "pickle":
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(mine, gathered, dst=0)
        if rank != 0:
            return None
        mine = [x for part in gathered for x in part]
    elif world > 1:
        from . import records
        table = records.gather_packed(dist, mine, device)
        if rank != 0:
            return None
        assert sorted(table.indices) == list(range(n_total))
        return table if results == "packed" else table.dicts()
    elif results == "packed":
        from . import records
        table = records.table_of([records.pack_many(mine)])
        assert sorted(table.indices) == list(range(n_total))
        return table
    by_index = dict(mine)
    assert sorted(by_index) == list(range(n_total))
    return [by_index[i] for i in range(n_total)]


# ----------------------------------------------------------------------------------------------------------------------
# Many recordings, several worker processes per GPU
# ----------------------------------------------------------------------------------------------------------------------
# The default (efficient) strategy decodes ONE stream token by token inside the ASR backend's own Python loop
# (transcribe.py:806 asserts batch 1): 2.7 ms of host time per token against a few hundred microseconds of GPU time, so one
# process leaves the GPU idle nine tenths of the time.  Recordings are independent units: W worker processes per GPU, each
# with its own copy of the model (whisper-base: 290 MB of 288 GB) and its own HIP queues, fill the GPU the way W batch
# streams would, with no change to the backend's loop and therefore to its output.  No collective anywhere: a job queue
# in, result dictionaries out.
def _many_worker(rank, n_workers, devices, load_model, my_audios, mine, options, barrier, out_queue, on_item, warmup,
                 streams=0, on_batch=None, barrier_timeout=600.0):
    """`mine` = the indices (into the caller's list) of this worker's recordings, `my_audios` = those recordings only."""
    import time
    import os
    # W processes with the default intra-op thread count each (= every core of the host) fight for cores (the CPU test of
    # this function ran 7x faster with a share per worker); the decode loop is one Python thread.  On the GPU this is not
    # what limits the scaling (the processes' small kernels serialise on the device)
    torch.set_num_threads(max(1, min(8, (os.cpu_count() or 8) // max(n_workers, 1))))
    dev = devices[rank % len(devices)]
    on_gpu = torch.device(dev).type == "cuda"      # (the CPU only ever appears in the host-logic tests)
    if on_gpu:
        torch.cuda.set_device(dev)
    from .transcribe import transcribe_batch, transcribe_timestamped
    model = load_model(dev)
    if warmup and mine and streams and streams > 1:
        # ONE recording through the B-stream driver (allocations, GEMM plans, arenas, the self-checks) -- not the worker's
        # whole batch: the workers' warm-ups would otherwise differ by whole batches at the barrier below
        if on_batch is not None:
            on_batch(mine[:1])
        transcribe_batch(model, my_audios[:1], max_streams=streams, **options)
    elif warmup and mine:                        # allocations, GEMM plans, the library's arenas: before the common start
        if on_item is not None:
            on_item(mine[0])
        transcribe_timestamped(model, my_audios[0], **options)
    if on_gpu:
        torch.cuda.synchronize(dev)
    if barrier is not None:
        barrier.wait(timeout=barrier_timeout)    # (BrokenBarrierError when the parent aborted it: a sibling died)
    t0 = time.perf_counter()
    res = []
    if streams and streams > 1:                  # the worker's recordings as decoder streams (streams.py)
        if on_batch is not None:
            on_batch(mine)
        res = list(zip(mine, transcribe_batch(model, my_audios, max_streams=streams, **options)))
    else:
        for i, audio in zip(mine, my_audios):
            if on_item is not None:
                on_item(i)
            res.append((i, transcribe_timestamped(model, audio, **options)))
    if on_gpu:
        torch.cuda.synchronize(dev)
    out_queue.put((rank, time.perf_counter() - t0, res))


def transcribe_many(load_model, audios, workers_per_gpu: int = 8, devices=None, on_item=None, warmup: bool = False,
                    return_timing: bool = False, streams: int = 0, on_batch=None, barrier_timeout: float = 600.0, **options):
    """transcribe_timestamped() of every recording in `audios` (1-D fp32 tensors / arrays at 16 kHz, or paths) on
    `workers_per_gpu` worker processes per GPU.  `load_model(device)` -> the model; it runs inside each worker and must
    be picklable (a module-level function), as must `on_item(index)` (called in the worker before item `index`).
    Recordings are dealt to the workers largest-first by length.  Returns the result dictionaries in the order of
    `audios` (with `return_timing`: also the slowest worker's seconds between the common start and its last result).
    `streams` > 1: every worker steps ITS recordings through the decoder together (transcribe_batch), so processes and
    decoder streams multiply -- a B-stream decoder loop is bound by its one Python thread, W processes run W of them;
    `on_batch(indices)` (picklable) is then called in the worker with its recordings' indices, in stream order."""
    import queue as queue_mod
    import torch.multiprocessing as mp
    from .naive import get_audio_tensor
    devices = list(devices) if devices is not None else [f"cuda:{k}" for k in range(torch.cuda.device_count())]
    assert devices, "transcribe_many needs at least one GPU"
    audios = [get_audio_tensor(a).cpu() for a in audios]
    n_workers = max(1, min(workers_per_gpu * len(devices), len(audios)))
    order = partition_units([int(a.numel()) for a in audios], n_workers)
    ctx = mp.get_context("spawn")
    barrier = ctx.Barrier(n_workers)
    queue = ctx.Queue()
    # every worker is sent ITS recordings only (spawn pickles the arguments: W x the whole list otherwise)
    procs = [ctx.Process(target=_many_worker, args=(r, n_workers, devices, load_model, [audios[i] for i in order[r]], order[r],
                                                    options, barrier, queue, on_item, warmup, streams, on_batch,
                                                    barrier_timeout))
             for r in range(n_workers)]
    for p in procs:
        p.daemon = True           # (a worker never outlives the process that asked for it)
        p.start()
    got, slowest = {}, 0.0
    failed = False
    try:
        for _ in procs:
            while True:
                try:
                    rank, seconds, res = queue.get(timeout=5.0)
                    break
                except queue_mod.Empty:        # nothing yet: is everybody still alive?
                    dead = [p for p in procs if p.exitcode not in (None, 0)]
                    if dead:
                        raise RuntimeError(f"transcribe_many: a worker died (exit code {dead[0].exitcode})")
            slowest = max(slowest, seconds)
            got.update(dict(res))
    except BaseException:
        # a worker that died before the barrier would leave its siblings waiting there for good: break the barrier and
        # stop everybody at once instead of joining blocked workers one by one
        failed = True
        barrier.abort()
        for p in procs:
            if p.is_alive():
                p.terminate()
        raise
    finally:
        for p in procs:
            p.join(timeout=5 if failed else 60)       # (success: a worker is tearing its HIP context down -- give it time)
            if p.is_alive():
                p.terminate()
    assert sorted(got) == list(range(len(audios)))
    results = [got[i] for i in range(len(audios))]
    return (results, slowest) if return_timing else results
