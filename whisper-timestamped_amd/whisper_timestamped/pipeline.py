"""The hot path's stages in flight: which HIP stream each kernel of a batch of 30 s chunks runs on, and what it waits for.

The reference runs the path one window at a time, in order
(/root/reference/whisper_timestamped/transcribe.py:1213-1214 log-mel, :1540-1569 cost, :1572-1581 + :1648-1652 DTW and
jumps, :1245/:1292 chosen-token log-probabilities).  On the MI355X a batch of chunks is ONE launch set per stage, and the
stages differ in what they need from the chip:

    stage     kernels                              bound by
    logmel    stft_mel, logmel_finalize, padding   VALU / LDS issue
    cost      rowmean, colnorm, fix00              HBM
    dtw       dtw_kernel                           latency (a chain of T + F dependent f64 steps, one workgroup per unit)
    logprob   logprob_gather                       HBM

``hilo`` puts the stages that cannot use the chip's bandwidth (logmel, dtw) on a HIGH-priority HIP stream and the
HBM-bound ones (cost, logprob) on a LOW-priority stream: the dispatcher places the high-priority workgroups first and the
bandwidth kernels fill what is left, so every HBM-bound kernel runs beside a compute- or latency-bound one.  Dependencies
travel as events (the DTW waits for its cost stage; a buffer set's next step waits for what still reads its buffers).
Several buffer sets in flight (``depth``) fill each other's gaps.  ``serial`` is one stream per buffer set, stages in order.

    StageSet          the streams + events of ONE buffer set in flight
    ChunkBatch        the device buffers of one batch of chunks (inputs, outputs, the KB-sized result record)
    HotPathPipeline   ``depth`` StageSets; ``submit(batch)`` queues the four stages and the result copy, nothing waits

``batched.BatchedAligner`` (the naive strategy's teacher-forced second pass) runs its stages through a StageSet;
``bench.py``'s timed region is ``HotPathPipeline.submit`` in a loop.  There is no CPU fallback (``_lib``).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field

import numpy as np
import torch

from . import _lib

STAGES = ("logmel", "cost", "dtw", "logprob")
# lane of a stage under `hilo`: "hi" = high-priority stream (cannot use the bandwidth), "lo" = low-priority (HBM-bound).
# "model" / "qk_rows" are the backend's GEMMs and the QK-row pass of the teacher-forced leg (batched.py).
LANE = {"logmel": "hi", "dtw": "hi", "cost": "lo", "logprob": "lo", "model": "lo", "qk_rows": "lo", "upload": "lo"}
SCHEDULES = ("serial", "hilo")
# `auto`: hilo where the latency-bound DTW leaves most of the chip idle (one workgroup per unit on a 256-CU chip) and the
# cost / DTW entry points are separate; measured on the 256-unit and the fused small-unit workloads, serial is as fast
# or faster there (profiles/r5m_schedules_secondary_workloads.jsonl).
HILO_MAX_UNITS = 128


def choose_schedule(schedule: str, n_units: int, fused_small_units: bool = False) -> str:
    if schedule != "auto":
        assert schedule in SCHEDULES, schedule
        return schedule
    return "hilo" if (n_units <= HILO_MAX_UNITS and not fused_small_units) else "serial"


_HIP = None


def _hip():
    global _HIP
    if _HIP is None:
        _HIP = ctypes.CDLL("libamdhip64.so")
    return _HIP


def stream_priorities():
    """{"low": least, "normal": 0, "high": greatest} of hipDeviceGetStreamPriorityRange (MI355X / ROCm 7.2: 1, 0, -1)."""
    lo, hi = ctypes.c_int(), ctypes.c_int()
    rc = _hip().hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi))
    if rc != 0:
        raise _lib.WtError(f"hipDeviceGetStreamPriorityRange failed: {rc}")
    return {"low": lo.value, "normal": 0, "high": hi.value}


def priority_stream(device, level: str):
    """A non-blocking HIP stream of the given priority level on `device`, as a torch stream object."""
    device = torch.device(device)
    with torch.cuda.device(device):
        st = ctypes.c_void_p()
        rc = _hip().hipStreamCreateWithPriority(ctypes.byref(st), ctypes.c_uint(1), ctypes.c_int(stream_priorities()[level]))
        if rc != 0:
            raise _lib.WtError(f"hipStreamCreateWithPriority failed: {rc}")
    return torch.cuda.ExternalStream(st.value, device=device)


class StageSet:
    """The streams and events of ONE buffer set in flight.

    ``run(stage, fn, after=..., events=...)`` launches ``fn(stream_handle)`` on the stage's stream once the LAST event
    recorded in this set for every stage named in ``after`` has fired, and every (event, stream) pair of ``events`` (the
    batch's own marks: what still reads the buffers this stage overwrites) -- a dependency on the same stream is ordered
    by the stream and costs no event wait -- then records the stage's event (``mark``: into the batch's own event).
    ``join(fn)`` queues ``fn()`` on the low-priority stream behind every stage launched since the last join."""

    def __init__(self, device, schedule: str = "hilo", streams=None, timeline: bool = False):
        self.device = torch.device(device)
        assert schedule in SCHEDULES, schedule
        self.schedule = schedule
        if streams is not None:
            self.hi, self.lo = streams
            self.owned = False
        elif schedule == "hilo":
            self.hi, self.lo = priority_stream(self.device, "high"), priority_stream(self.device, "low")
            self.owned = True
        else:
            self.hi = self.lo = torch.cuda.Stream(device=self.device)
            self.owned = True
        self.events = {}                             # stage -> (last event recorded for it in this set, its stream)
        self._own = {}
        self._since_join = []
        self.timeline = [] if timeline else None     # [(stage, start_event, end_event)] when asked for
        self._timing_pool = []
        self._last_mark = None

    @classmethod
    def on_current_stream(cls, device, timeline=False):
        """Everything on the caller's current stream, in order (the single-stream pass the stage times come from)."""
        cur = torch.cuda.current_stream(torch.device(device))
        return cls(device, "serial", streams=(cur, cur), timeline=timeline)

    def start_timeline(self, on: bool = True):
        """Collect (stage, start_event, end_event) for every stage run from now on (``on=False``: stop, drop what was kept).
        The timing events of the entries dropped here are recycled (their times must have been read by now)."""
        if self.timeline:
            seen = set()
            for _, ev0, ev1 in self.timeline:
                for ev in (ev0, ev1):
                    if id(ev) not in seen:
                        seen.add(id(ev))
                        self._timing_pool.append(ev)
        self.timeline = [] if on else None
        self._last_mark = None

    def _timing_event(self):
        return self._timing_pool.pop() if self._timing_pool else torch.cuda.Event(enable_timing=True)

    def stream_of(self, stage):
        return self.hi if LANE.get(stage, "lo") == "hi" else self.lo

    def wait_for(self, st, stages=(), events=()):
        for dep in stages:
            ev = self.events.get(dep)
            if ev is not None and ev[1] is not st:
                st.wait_event(ev[0])
        for ev in events:
            if ev[1] is not None and ev[1] is not st:
                st.wait_event(ev[0])

    def run(self, stage, fn, after=(), events=(), mark=None):
        st = self.stream_of(stage)
        self.wait_for(st, after, events)
        if self.timeline is not None:
            # back to back on one stream the end mark of a stage IS the start mark of the next (one event less in the queue)
            if self._last_mark is not None and self._last_mark[1] is st:
                start = self._last_mark[0]
            else:
                start = self._timing_event()
                start.record(st)
        fn(st.cuda_stream)
        if mark is not None:
            ev = mark[0]
            mark[1] = st
        else:
            ev = self._own.get((stage, id(st)))
            if ev is None:
                ev = self._own[(stage, id(st))] = torch.cuda.Event()
        ev.record(st)
        self.events[stage] = (ev, st)
        self._since_join.append(stage)
        if self.timeline is not None:
            end = self._timing_event()
            end.record(st)
            self.timeline.append((stage, start, end))
            self._last_mark = (end, st)
        return st

    def drain_into(self, st):
        """`st` waits for every stage launched in this set since the last join."""
        self.wait_for(st, self._since_join)
        self._since_join = []
        self._last_mark = None

    def join(self, fn, mark=None):
        """``fn()`` on the low-priority stream (made current for the call) behind every stage launched since the last join."""
        self.drain_into(self.lo)
        with torch.cuda.stream(self.lo):
            fn()
        if mark is not None:
            mark[0].record(self.lo)
            mark[1] = self.lo
        return self.lo

    def synchronize(self):
        self.hi.synchronize()
        if self.lo is not self.hi:
            self.lo.synchronize()

    def release(self):
        """Free the library's scratch arenas of this set's streams (their work must have completed)."""
        if self.owned:
            for s in {id(self.hi): self.hi, id(self.lo): self.lo}.values():
                _lib.release_stream(s)


def result_record(n_jumps: int, n_logprob: int, device):
    """The KB-sized results of a step live in ONE device record (jumps, then the log-probabilities) so that a single
    asynchronous copy brings them to the host (and a single message carries them to rank 0)."""
    rec = torch.empty(n_jumps + n_logprob, dtype=torch.int32, device=device)
    host = torch.empty(n_jumps + n_logprob, dtype=torch.int32).pin_memory()
    return rec, host


@dataclass
class ChunkBatch:
    """Device buffers of one batch of 30 s chunks on the hot path.  Inputs are the caller's (PCM, the alignment heads' QK
    rows, the decoder's logits rows and the tokens chosen from them); outputs are written by the four stages.  Units are
    described by ``wt_seg_desc`` records (include/wtalign.h) laid out by ``_lib.layout_outputs``."""
    pcm: torch.Tensor                   # (n_chunks, 480000) f32
    n_valid: torch.Tensor               # (n_chunks,) i32: real samples per chunk (the rest is silence / padding)
    mel_fb: torch.Tensor                # (n_mels, 201) f32
    qk: torch.Tensor                    # the selected heads' QK rows, f32 or f16; units address it through descs
    descs: np.ndarray                   # wt_seg_desc[n_units] (host)
    descs_dev: torch.Tensor             # the same bytes on the device
    head_idx: torch.Tensor              # (A,) i32
    logits: torch.Tensor                # (n_rows, V) f32
    tokens: torch.Tensor                # (n_rows,) i32
    mel: torch.Tensor                   # out (n_chunks, n_mels, 3000) f32
    gmax: torch.Tensor                  # out (n_chunks,) f32
    pad: torch.Tensor                   # out (n_chunks,) i32: find_start_padding of every window (-1 = None)
    cost: torch.Tensor                  # out: the units' cost matrices (f32, at descs' cost_offset)
    result: torch.Tensor                # out: int32 record = jumps | log-probabilities (f32 bits)
    host_result: torch.Tensor           # pinned host mirror of `result`
    n_jumps: int
    medfilt_width: int = 9
    qk_scale: float = 1.0
    fused_small_units: bool = False     # True: ONE wt_align_batch_v3 as the cost stage (small units through the fused tail
    #                                     kernel), no separate dtw stage
    row0: int = 0                       # (sub-batches) first logits row / first chunk of this view in the parent
    chunk0: int = 0
    parent: object = None
    _n_frames: int = 3000
    _calls: dict = field(default=None, repr=False)
    # [event, stream it was last recorded on]: the batch's last log-mel (writes `mel` in two passes), its last DTW (reads
    # `cost`) and its last result copy (reads `result`)
    mel_done: list = field(default_factory=lambda: [torch.cuda.Event(), None], repr=False)
    dtw_done: list = field(default_factory=lambda: [torch.cuda.Event(), None], repr=False)
    copied: list = field(default_factory=lambda: [torch.cuda.Event(), None], repr=False)

    # ---- views
    @property
    def n_chunks(self):
        return self.pcm.shape[0]

    @property
    def n_units(self):
        return len(self.descs)

    @property
    def n_rows(self):
        return self.logits.shape[0]

    @property
    def jumps(self):
        return self.result[:self.n_jumps]

    @property
    def logprob(self):
        return self.result[self.n_jumps:].view(torch.float32)

    @property
    def host_jumps(self):
        return self.host_result[:self.n_jumps]

    @property
    def host_logprob(self):
        return self.host_result[self.n_jumps:].view(torch.float32)

    @property
    def device(self):
        return self.pcm.device

    def twin(self):
        """Another buffer set over the SAME inputs: its own outputs and result record (two batches in flight)."""
        dev = self.device
        rec, host = result_record(self.n_jumps, self.result.numel() - self.n_jumps, dev)
        return ChunkBatch(self.pcm, self.n_valid, self.mel_fb, self.qk, self.descs, self.descs_dev, self.head_idx, self.logits,
                          self.tokens, torch.empty_like(self.mel), torch.empty_like(self.gmax), torch.empty_like(self.pad),
                          torch.empty_like(self.cost), rec, host, self.n_jumps, self.medfilt_width, self.qk_scale,
                          self.fused_small_units)

    def split(self, parts: int, rows_per_chunk: int):
        """`parts` sub-batches over the same buffers (chunk ranges; needs one unit per chunk, in chunk order, and
        `rows_per_chunk` logits rows per chunk): each is launched as a batch of its own, results land in this record."""
        n = self.n_chunks
        assert self.n_units == n and self.n_rows == n * rows_per_chunk, "split(): one unit and a fixed number of rows per chunk"
        assert not self.fused_small_units
        out = []
        sz = _lib.SEG_DTYPE.itemsize
        for p in range(parts):
            b0, b1 = n * p // parts, n * (p + 1) // parts
            if b0 == b1:
                continue
            r0, r1 = b0 * rows_per_chunk, b1 * rows_per_chunk
            out.append(ChunkBatch(self.pcm[b0:b1], self.n_valid[b0:b1], self.mel_fb, self.qk, self.descs[b0:b1],
                                  self.descs_dev[b0 * sz:b1 * sz], self.head_idx, self.logits[r0:r1], self.tokens[r0:r1],
                                  self.mel[b0:b1], self.gmax[b0:b1], self.pad[b0:b1], self.cost, self.result, self.host_result,
                                  self.n_jumps, self.medfilt_width, self.qk_scale, False, row0=r0, chunk0=b0, parent=self))
        return out

    # ---- the four stages (each: one entry point of libwtalign.so on the given stream handle)
    def stage_calls(self):
        if self._calls is not None:
            return self._calls
        L = _lib.load()
        chk = _lib._check
        n, n_units, n_rows = self.n_chunks, self.n_units, self.n_rows
        n_mels, V = self.mel_fb.shape[0], self.logits.shape[1]
        assert self.pcm.is_contiguous() and self.mel.is_contiguous() and self.logits.stride(1) == 1
        for t in (self.pcm, self.qk, self.logits, self.tokens, self.mel, self.cost, self.result, self.descs_dev):
            _lib._need_cuda(t, "ChunkBatch buffers")
        _lib.same_device(self.pcm, self.n_valid, self.mel_fb, self.qk, self.descs_dev, self.head_idx, self.logits, self.tokens,
                         self.mel, self.gmax, self.pad, self.cost, self.result)
        qk_dt = {torch.float32: _lib.WT_DTYPE_F32, torch.float16: _lib.WT_DTYPE_F16}[self.qk.dtype]
        pcm, nv, fb, mel, gmax, pad = (t.data_ptr() for t in (self.pcm, self.n_valid, self.mel_fb, self.mel, self.gmax, self.pad))
        qk, dh, dd, hi = self.qk.data_ptr(), self.descs.ctypes.data, self.descs_dev.data_ptr(), self.head_idx.data_ptr()
        A, w9, sc = self.head_idx.numel(), int(self.medfilt_width), float(self.qk_scale)
        cost, jumps = self.cost.data_ptr(), self.result.data_ptr()
        logits, toks = self.logits.data_ptr(), self.tokens.data_ptr()
        lp = self.result.data_ptr() + 4 * (self.n_jumps + self.row0)
        n_samples, n_frames, ldl = self.pcm.shape[1], self._n_frames, self.logits.stride(0)

        def logmel(st):
            chk(L.wt_logmel_pad_batch(pcm, n, n_samples, nv, fb, n_mels, n_frames, mel, gmax, pad, st), "wt_logmel_pad_batch")

        if self.fused_small_units:
            def cost_stage(st):
                chk(L.wt_align_batch_v3(qk, qk_dt, dh, dd, n_units, hi, A, w9, sc, cost, jumps, 0, 0, 0, 0, 0, st), "wt_align_batch_v3")

            def dtw(st):
                pass
        else:
            def cost_stage(st):
                chk(L.wt_cost_batch(qk, qk_dt, dh, dd, n_units, hi, A, w9, sc, cost, st), "wt_cost_batch")

            def dtw(st):
                chk(L.wt_dtw_batch(cost, dh, dd, n_units, jumps, 0, 0, 0, 0, st), "wt_dtw_batch")

        def logprob(st):
            chk(L.wt_logprob_gather_batch(logits, 0, ldl, n_rows, V, toks, 0, 0, lp, st), "wt_logprob_gather_batch")

        self._calls = dict(logmel=logmel, cost=cost_stage, dtw=dtw, logprob=logprob)
        return self._calls

    def fetch(self):
        """Queue the ONE device->host copy of the result record on the current stream."""
        self.host_result.copy_(self.result, non_blocking=True)


class HotPathPipeline:
    """``depth`` buffer sets in flight over the hot path's four stages.

    ``submit(batch)`` queues log-mel (+ padding index), cost, DTW (+ backtrack, jumps), log-probability gather and the
    asynchronous copy of the result record for one ChunkBatch and returns the stream the copy is queued on; nothing
    waits.  The k-th submit uses StageSet k % depth: give every set in flight its own ChunkBatch (``batch.twin()``) --
    a set's next step waits for what still reads its buffers, two sets never wait for each other.  ``sub_batches`` > 1
    launches a batch as that many chunk ranges, round-robin over the sets (large batches: the DTW and the log-mel of one
    range run beside the HBM-bound kernels of another; the copy follows the last range)."""

    def __init__(self, device, depth: int = 2, schedule: str = "auto", n_units: int | None = None, fused_small_units: bool = False,
                 timeline: bool = False, sub_batches: int = 1, rows_per_chunk: int | None = None):
        self.device = torch.device(device)
        _lib.require_gpu(self.device, "the hot-path pipeline")
        self.schedule = choose_schedule(schedule, n_units if n_units is not None else 0, fused_small_units)
        self.depth = int(depth)
        self.sub_batches, self.rows_per_chunk = int(sub_batches), rows_per_chunk
        if self.depth == 1 and self.schedule == "serial":
            self.sets = [StageSet.on_current_stream(self.device, timeline=timeline)]
        else:
            self.sets = [StageSet(self.device, self.schedule, timeline=timeline) for _ in range(self.depth)]
        self._k = 0
        self._parts = {}

    def describe(self):
        return {"class": "whisper_timestamped.pipeline.HotPathPipeline", "schedule": self.schedule, "batches_in_flight": self.depth,
                "sub_batches": self.sub_batches,
                "streams": {s: LANE[s] if self.schedule == "hilo" else "one stream per buffer set" for s in STAGES}}

    @staticmethod
    def _stages(s: StageSet, batch: ChunkBatch, record: ChunkBatch):
        """The four stages of `batch` on the streams of `s`.  What a stage must wait for beyond its own step is kept with the
        buffers, not with the streams: the batch's last DTW still reads the cost buffer the cost stage overwrites; the last
        copy of `record` (the batch itself, or the parent of a chunk range) still reads the record the DTW and the gather
        write."""
        calls = batch.stage_calls()
        copied = (record.copied,)
        # (the front end is two passes over `mel` -- raw log-mel, then the clamp in place: a second submit of the same
        #  batch on another stream set must not start while the first is between them)
        s.run("logmel", calls["logmel"], events=(batch.mel_done,), mark=batch.mel_done)
        if batch.fused_small_units:
            s.run("cost", calls["cost"], events=copied)
        else:
            s.run("cost", calls["cost"], events=(batch.dtw_done,))
            s.run("dtw", calls["dtw"], after=("cost",), events=copied, mark=batch.dtw_done)
        s.run("logprob", calls["logprob"], events=copied)

    def submit(self, batch: ChunkBatch):
        if self.sub_batches > 1:
            return self._submit_parts(batch)
        s = self.sets[self._k % self.depth]
        self._k += 1
        self._stages(s, batch, batch)
        return s.join(batch.fetch, mark=batch.copied)

    def _submit_parts(self, batch: ChunkBatch):
        parts = self._parts.get(id(batch))
        if parts is None:
            parts = self._parts[id(batch)] = batch.split(self.sub_batches, self.rows_per_chunk)
        used = []
        for part in parts:                           # every range writes its own slice of the parent's buffers
            s = self.sets[self._k % self.depth]
            self._k += 1
            self._stages(s, part, batch)
            if s not in used:
                used.append(s)
        last = used[-1]
        for s in used[:-1]:                          # the record is copied once, behind all of them
            s.drain_into(last.lo)
        return last.join(batch.fetch, mark=batch.copied)

    def synchronize(self):
        for s in self.sets:
            s.synchronize()

    def release(self):
        for s in self.sets:
            s.release()
