"""The efficient strategy for B INDEPENDENT recordings at once: B decoder streams per decoder op.

The reference's efficient strategy (/root/reference/whisper_timestamped/transcribe.py:359-1001, "T.py" below) hooks
openai-whisper's own ``transcribe()`` loop, which decodes ONE stream token by token (T.py:806 asserts a batch of one):
on an MI355X that is ~2.7 ms of host Python per token around a few hundred microseconds of GPU work.
Recordings are independent units, so here B of them step through the decoder TOGETHER:

  * a lock-step window driver: every round each active stream contributes its next 30 s window (its own seek, its own
    prompt); streams whose initial token rows have the same length share ONE batched decoder loop -- the backend's own
    ``DecodingTask._main_loop`` (KV cache, logit filters, sampler), entered with one row of initial tokens PER STREAM
    instead of ``DecodingTask.run``'s one row repeated.  The seek / prompt / no-speech / segment-splitting logic around
    it is openai-whisper's ``transcribe()`` loop restated per stream (whisper/transcribe.py; third-party, absent from
    the build image: SURVEY.md Appendix C) for the option domain of the efficient strategy: greedy or single-temperature
    sampling, no temperature fallback, no beam (those go to the naive strategy: T.py:243-252), no ``vad``;
  * the data plane of the decode-time hooks for all streams of a decoder call: ONE ``wt_qk_rows_streams`` launch per
    token writes the alignment heads' QK rows of every stream into its block of a (B, A_sel, sample_len + 1, 1500) ring,
    ONE ``wt_logprob_digest_streams`` launch per token takes from the (B, V) rows the sampler has just filtered what the
    hook state machine can still ask of them (the sampled token's log-probability, the argmax, max / log-sum-exp, the
    raw logits of <|endoftext|> / <|notimestamps|> and of every timestamp token: 32 B + 6 KB per row -- the rows
    themselves are not kept; round 4 kept them in a (B, 449, V) ring, 93 MB per stream), ONE host copy per decoder loop
    brings every stream's digests, ONE ``wt_find_start_padding_batch`` serves all windows of a round;
  * per stream an unmodified ``EfficientSession`` (efficient.py: the reference's hook state machine, decision by
    decision) that is fed the RECORDED decoder calls of its stream -- the same methods in the same order as the live
    hooks would call them -- and whose alignment units go into ONE ``AlignmentBatch`` per window set (the sink).

Ordering, which is what makes the rings safe to recycle: the first decoder call of a window (its prompt) is what
flushes the previous window's last segment and closes that window (T.py:824, Appendix B of SURVEY.md), and the prompt
is known to the driver BEFORE the batched loop runs.  So a round is: mel windows -> padding detector -> every active
stream's prompt call replayed (previous window flushed and closed, its units queued) -> ONE alignment launch set ->
the batched decoder loops (which overwrite the rings: the kernels that read the old rows are already queued on the
same HIP stream) -> the remaining recorded calls replayed.

Which streams share a decoder loop: those whose initial-token rows have the same LENGTH (the backend's decoder has no
padding mask; padding a shorter prompt would move every position and change the result).  Recordings of <= 30 s
(BASELINE configs[1]), first windows, unconditioned decoding (condition_on_previous_text=False) and saturated prompts
(223 tokens, after two or three dense windows) all meet in one loop per round; with the reference's default
conditioning the second / third windows of a recording carry prompts of their own lengths and form small loops of
their own -- correct, less batched.  A recording that is finished hands its ring block to the next one (continuous
admission), so short and long recordings mix without the long ones ending up alone.

Results: for a given stream the host logic is byte for byte the B = 1 code; the numerics differ from a B = 1 run only
through the batch size of the backend's GEMMs (tests: B = 8 equals eight B = 1 runs word for word, time for time).
"""
from __future__ import annotations

import logging
import threading

import numpy as np
import torch

from . import _lib, backend, efficient
from .alignment import AlignmentBatch, default_workspace, head_pairs
from .capture import layer_head_slots
from .efficient import EfficientSession
from .words import HOP_LENGTH, N_FRAMES, SAMPLE_RATE

logger = logging.getLogger("whisper_timestamped")

N_SAMPLES = 30 * SAMPLE_RATE
# test / instrumentation seam: called with the RECORDING indices (positions in `audios`) of the streams of a batched decoder
# loop, in row order, right before it starts
ON_GROUP_DECODE = None
LAST_RUN = {}
# (Round 5 measured "bucket admission" -- a stream whose prompt length nobody shares sits one round out -- and found no
#  gain: prompt lengths are spread over ~120 values and holding a stream does not change its length; removed in round 6,
#  profiles/r5f_bench_driver_command.json `ragged_bucket_admission` keeps the record.)
PAUSE_GC = True               # pause Python's cyclic garbage collector while a batch decodes (see _run_streams)
GC_EVERY_ROUNDS = 8           # ... with a young-generation pass every this many rounds (a long batch must not hoard cycles)


class paused_gc:
    """`with paused_gc():` -- the cyclic collector off for the block (when PAUSE_GC and it was on), back on afterwards and
    run once.  NB ``gc.disable()`` is PROCESS-global: other threads of a host application are affected for the duration of
    the batch (set ``streams.PAUSE_GC = False`` there).  Nested / concurrent blocks are counted: the collector comes back
    when the LAST one exits."""
    _depth = 0
    _was_enabled = False
    _lock = threading.Lock()

    def __enter__(self):
        import gc
        self.mine = False
        if PAUSE_GC:
            with paused_gc._lock:
                if paused_gc._depth == 0:
                    paused_gc._was_enabled = gc.isenabled()
                    if paused_gc._was_enabled:
                        gc.disable()
                paused_gc._depth += 1
                self.mine = True
        return self

    def tick(self, rounds):
        """Called once per round of the driver: a cheap young-generation collection every GC_EVERY_ROUNDS rounds."""
        if self.mine and GC_EVERY_ROUNDS and rounds % GC_EVERY_ROUNDS == 0:
            import gc
            gc.collect(0)

    def __exit__(self, *exc):
        if self.mine:
            import gc
            with paused_gc._lock:
                paused_gc._depth -= 1
                last = paused_gc._depth == 0 and paused_gc._was_enabled
                if last:
                    gc.enable()
            if last:
                gc.collect()


# how often a stream's session asked for something other than the sampled token's log-probability (the reference's
# fallbacks for a stuck decoder): tests / diagnostics
FALLBACK_READS = {"argmax": 0, "argmax_over_later_timestamps": 0, "logprob_of_another_token": 0, "from_the_last_row_kept_whole": 0}


# ----------------------------------------------------------------------------------------------------------------------
# device rings of B streams
# ----------------------------------------------------------------------------------------------------------------------
class _QKView:
    """What an EfficientSession needs of a QKCaptureRing, on one stream's block of the shared ring."""

    def __init__(self, buf):
        self.buf = buf                      # (A_sel, capacity, n_ctx)
        self.device = buf.device

    def rows(self, rows):
        rows = list(rows)
        if rows and rows == list(range(rows[0], rows[0] + len(rows))):
            return self.buf[:, rows[0]:rows[0] + len(rows)]
        return self.buf.index_select(1, torch.tensor(rows, dtype=torch.long, device=self.device))


class _LogitsView:
    """What an EfficientSession needs of a LogitsRing, for one stream of the B-stream driver.  No logits row is kept: when
    a decoder call's rows are final, ONE launch for all streams of the call (wt_logprob_digest_streams) takes from each
    row what the hook state machine can still ask of it -- the sampled token's log-probability (T.py:735), the row's
    argmax (T.py:508,729,879), its max / log-sum-exp and the raw logits of <|endoftext|>, <|notimestamps|> and of every
    timestamp token (T.py:535 and the fallback tokens of a stuck decoder) -- 32 bytes + 6 KB per row instead of 207 KB.
    The driver hands over the digests of the stream's window as ONE host array per decoder loop (`host`), so `gather`
    and `argmax` are host reads; only a timestamp token that was not the sampled one goes back to the device slice."""

    def __init__(self, rings, block):
        self.rings, self.block = rings, block
        self.n = 0
        self.host = None                    # (n_calls, 8) fp32: this window's digest records, on the host
        self.sampled = None                 # int64[n_calls]: the token sampled at each call
        self.full_row = None                # index of the ONE row kept whole (rings.last_row): the loop's final call

    def reset(self):
        self.n = 0

    def append(self, row):
        self.n += 1

    def __len__(self):
        return self.n

    def window(self, host, sampled, full_row=None):
        self.host, self.sampled = host, np.asarray(sampled, dtype=np.int64)
        self.full_row = full_row

    def argmax(self, row, lo=0):
        if row < 0:
            row += self.n
        if lo == 0:
            FALLBACK_READS["argmax"] += 1
            return int(self.host[row, 3:4].view(np.int32)[0])
        FALLBACK_READS["argmax_over_later_timestamps"] += 1
        r = self.rings
        assert lo >= r.slice_begin, f"argmax(row, lo={lo}): only the timestamp slice of a row is kept"
        return int(torch.argmax(r.slice[self.block, row, lo - r.slice_begin:]).item()) + lo

    def _logprob_of(self, row, tok):
        """log_softmax(row)[tok] for a token other than the sampled one (the reference's fallbacks for a stuck decoder)."""
        d, r = self.host[row], self.rings
        FALLBACK_READS["logprob_of_another_token"] += 1
        if tok == int(d[3:4].view(np.int32)[0]):
            x = d[1]
        elif tok in r.aux_tokens:
            x = d[4 + r.aux_tokens.index(tok)]
        elif tok >= r.slice_begin:
            x = np.float32(r.slice[self.block, row, tok - r.slice_begin].item())
        elif self.full_row is not None and row == self.full_row:
            # a stuck decoder's fallback token taken from the NEXT window's prompt (T.py:498-503) can be any text token;
            # it is only ever asked of the window's last row when the decoding limit was hit: the loop's final call,
            # whose rows are kept whole (one per stream)
            FALLBACK_READS["from_the_last_row_kept_whole"] += 1
            x = np.float32(r.last_row[self.block, tok].item())
        else:
            raise _lib.WtError(f"streams: the log-probability of token {tok} at step {row} was asked for, which is neither the sampled "
                               f"token, the most likely one, a special token nor a timestamp, nor of the loop's last call: not kept "
                               f"by the B-stream driver")
        return np.float32(np.float32(x - d[1]) - d[2])

    def gather(self, tokens):
        n = len(tokens)
        assert n <= self.n and self.host is not None and n <= len(self.host)
        tok = np.asarray(tokens, dtype=np.int64)
        out = self.host[:n, 0].copy()
        for i in np.nonzero(tok != self.sampled[:n])[0]:
            out[i] = self._logprob_of(int(i), int(tok[i]))
        return torch.from_numpy(out)


class StreamRings:
    """(B, A_sel, capacity, n_ctx) QK logits of the alignment heads + per decoder call and stream a 32-byte digest of the
    filtered logits row and its timestamp slice ((B, capacity, 8) and (B, capacity, V - timestamp_begin) fp32).
    capacity = the decoder calls a window can take (sample_len + 1)."""

    def __init__(self, model, alignment_heads, hooked_blocks, n_streams, dtype, sample_len=None, tokenizer=None):
        dev = model.device
        dims = model.dims
        self.device = dev
        calls = min(dims.n_text_ctx, int(sample_len or dims.n_text_ctx // 2) + 1)
        self.n_streams, self.capacity, self.n_ctx = n_streams, calls, dims.n_audio_ctx
        per_layer, self.n_slots = layer_head_slots(head_pairs(alignment_heads), len(hooked_blocks), dims.n_text_head)
        self.used = [l for l, (h, _) in enumerate(per_layer) if h]
        sel = [(i, h, s) for i, l in enumerate(self.used) for h, s in zip(*per_layer[l])]
        self.sel = tuple(torch.tensor([x[k] for x in sel], dtype=torch.int32, device=dev) for k in range(3))
        self.n_sel = len(sel)
        self.qk = torch.zeros((n_streams, max(self.n_slots, 1), self.capacity, self.n_ctx), dtype=dtype, device=dev)
        tk = tokenizer if tokenizer is not None else backend.get_tokenizer(model, task="transcribe", language="en")
        self.slice_begin = int(tk.timestamp_begin)
        self.aux_tokens = [int(tk.eot)] + ([int(tk.no_timestamps)] if tk.no_timestamps is not None else [])
        self.digest = torch.zeros((n_streams, self.capacity, _lib.DIGEST_WORDS), dtype=torch.float32, device=dev)
        self.slice = torch.zeros((n_streams, self.capacity, dims.n_vocab - self.slice_begin), dtype=torch.float32, device=dev)
        # the filtered logits row of each stream's LAST decoder call of its current loop, whole (V floats per stream)
        self.last_row = torch.zeros((n_streams, dims.n_vocab), dtype=torch.float32, device=dev)
        self._dt = {torch.float32: _lib.WT_DTYPE_F32, torch.float16: _lib.WT_DTYPE_F16}[dtype]
        self._lib = _lib.load()
        import ctypes as C
        self._qp, self._kp = (C.c_void_p * len(self.used))(), (C.c_void_p * len(self.used))()

    @staticmethod
    def bytes_per_stream(model, alignment_heads, hooked_blocks, dtype, sample_len, tokenizer):
        dims = model.dims
        tokenizer = tokenizer if tokenizer is not None else backend.get_tokenizer(model, task="transcribe", language="en")
        calls = min(dims.n_text_ctx, int(sample_len or dims.n_text_ctx // 2) + 1)
        _, n_slots = layer_head_slots(head_pairs(alignment_heads), len(hooked_blocks), dims.n_text_head)
        item = 2 if dtype == torch.float16 else 4
        return calls * (max(n_slots, 1) * dims.n_audio_ctx * item + 4 * (_lib.DIGEST_WORDS + dims.n_vocab - int(tokenizer.timestamp_begin))) \
            + 4 * dims.n_vocab

    def write_digest(self, rows, tokens, ring_index, step):
        """rows (g, V): a decoder call's last-position logits as the sampler left them; tokens (g,): what it sampled."""
        if rows.dtype != torch.float32:      # a backend whose decoder hands out half-precision logits (the ring copy of round 4 converted)
            rows = rows.float()
        _lib.logprob_digest_streams(rows, tokens, ring_index, self.digest, self.slice, step, self.aux_tokens, self.slice_begin)

    def keep_last_rows(self, rows, ring_index_long):
        """The loop's final call: its rows whole, one per stream (what a prompt-derived fallback token is read from)."""
        self.last_row.index_copy_(0, ring_index_long, rows.float())

    def write_qk(self, q_layers, k_layers, ring_index, row):
        """The LAST query row of every selected head, for every stream of the call: q (g, n_q, D), K (g, n_ctx, D) per
        used layer; ring_index: device int32[g] = each batch entry's stream block."""
        if self.n_sel == 0:
            return
        q0, k0 = q_layers[self.used[0]], k_layers[self.used[0]]
        g, n_q, D = q0.shape
        assert q0.stride(2) == 1 and q0.stride(1) == D and k0.shape == (g, self.n_ctx, D) and k0.stride(1) == D and \
            k0.stride(2) == 1 and k0.dtype == q0.dtype and D % 64 == 0, (q0.shape, k0.shape, q0.dtype)
        last = (n_q - 1) * D * q0.element_size()
        for i, l in enumerate(self.used):
            assert q_layers[l].shape == q0.shape and q_layers[l].stride() == q0.stride() and k_layers[l].stride() == k0.stride()
            self._qp[i] = q_layers[l].data_ptr() + last
            self._kp[i] = k_layers[l].data_ptr()
        dt = _lib.WT_DTYPE_F32 if q0.dtype == torch.float32 else _lib.WT_DTYPE_F16
        sl, sh, ss = self.sel
        with _lib.on_device(self.qk) as st:
            rc = self._lib.wt_qk_rows_streams(self._qp, self._kp, len(self.used), dt, g, 1, q0.stride(0), k0.stride(0), self.n_ctx,
                                              D, 64, 64.0 ** -0.25, sl.data_ptr(), sh.data_ptr(), ss.data_ptr(), self.n_sel,
                                              ring_index.data_ptr(), self.qk.data_ptr(), self._dt, self.qk.stride(0),
                                              self.capacity, int(row), st)
        _lib._check(rc, "wt_qk_rows_streams")


# ----------------------------------------------------------------------------------------------------------------------
# one AlignmentBatch per window set
# ----------------------------------------------------------------------------------------------------------------------
class _Sink:
    def __init__(self, workspace):
        self.workspace = workspace
        self.pending = []                   # (unit, placeholder words, padding handle) of every stream, in arrival order
        self.in_flight = []
        self.launch_sets = 0

    def take(self, queued):
        self.pending.extend(queued)

    def launch(self):
        """ONE launch set for everything queued since the last one; then the PREVIOUS set's record is read (its copy was
        queued a whole decoder loop ago)."""
        previous, self.in_flight = self.in_flight, []
        if self.pending:
            batch = AlignmentBatch(workspace=self.workspace)
            for unit, _, handle in self.pending:
                if handle is not None:
                    sp = int(handle[1].wait()[0])
                    efficient.set_padding(unit, None if sp < 0 else sp)
                batch.add(unit)
            batch.launch().fetch()
            self.launch_sets += 1
            self.in_flight.append((batch, [ws for _, ws, _ in self.pending]))
            self.pending = []
        EfficientSession._collect(previous)

    def resolve(self):
        self.launch()
        pending, self.in_flight = self.in_flight, []
        EfficientSession._collect(pending)


class _SliceOfCopy:
    """handle.wait()[0] for stream j of a shared asynchronous host copy (the padding indices of a whole round)."""

    def __init__(self, shared, j):
        self.shared, self.j = shared, j

    def wait(self):
        return self.shared.wait()[self.j:self.j + 1]


# ----------------------------------------------------------------------------------------------------------------------
# the backend's timestamp rules, all rows at once
# ----------------------------------------------------------------------------------------------------------------------
# openai-whisper's ApplyTimestampRules.apply (whisper/decoding.py) walks the batch ROW BY ROW on the host: a .tolist()
# of the row's sampled tokens, up to three slice assignments, a logsumexp / max / comparison -- the comparison being a
# device -> host synchronisation per row.  With one stream that is noise; with 32 streams it was HALF of the wall time
# of a decoder loop (profiles/r4d_profile_streams.txt: 0.52 of 1.06 s).  Same rules, same masks, computed for all rows
# on the device with no host read: the driver swaps this object for the backend's inside the task it drives
# (VECTORIZED_TIMESTAMP_RULES = False keeps the backend's own).  Held against the backend's class on random token
# histories (tests/test_streams_host.py): identical logits.
VECTORIZED_TIMESTAMP_RULES = True


class BatchedTimestampRules:
    def __init__(self, tokenizer, sample_begin, max_initial_timestamp_index):
        self.tokenizer, self.sample_begin = tokenizer, sample_begin
        self.max_initial_timestamp_index = max_initial_timestamp_index
        self._col = None

    @classmethod
    def like(cls, rule):
        return cls(rule.tokenizer, rule.sample_begin, rule.max_initial_timestamp_index)

    def apply(self, logits, tokens):
        tk = self.tokenizer
        ts0, eot, V = tk.timestamp_begin, tk.eot, logits.shape[-1]
        if tk.no_timestamps is not None:
            logits[:, tk.no_timestamps] = -np.inf
        if self._col is None or self._col.device != logits.device or self._col.numel() != V:
            self._col = torch.arange(V, device=logits.device)
        col = self._col[None]
        sampled = tokens[:, self.sample_begin:].to(logits.device)
        B, n = sampled.shape
        if n >= 1:
            is_ts = sampled >= ts0
            last = is_ts[:, -1]
            penult = is_ts[:, -2] if n >= 2 else torch.ones(B, dtype=torch.bool, device=logits.device)
            # timestamps come in pairs, except right before eot
            mask = ((last & penult)[:, None] & (col >= ts0)) | ((last & ~penult)[:, None] & (col < eot))
            # timestamps never decrease: below the LAST timestamp of the row (+ 1 unless it opens a pair)
            pos = torch.arange(n, device=logits.device)[None]
            where_last = torch.where(is_ts, pos, torch.full_like(pos, -1)).max(dim=1).values      # -1: no timestamp yet
            any_ts = where_last >= 0
            last_ts = sampled.gather(1, where_last.clamp(min=0)[:, None])[:, 0]
            bound = last_ts + torch.where(last & ~penult, 0, 1)
            mask |= any_ts[:, None] & (col >= ts0) & (col < bound[:, None])
            logits.masked_fill_(mask, -np.inf)
        if tokens.shape[1] == self.sample_begin:
            logits[:, :ts0] = -np.inf                  # the first sampled token is a timestamp
            if self.max_initial_timestamp_index is not None:
                logits[:, ts0 + self.max_initial_timestamp_index + 1:] = -np.inf
        # the total timestamp mass above any text token -> a timestamp it is
        logprobs = torch.nn.functional.log_softmax(logits.float(), dim=-1)
        take_ts = logprobs[:, ts0:].logsumexp(dim=-1) > logprobs[:, :ts0].max(dim=-1).values
        logits[:, :ts0].masked_fill_(take_ts[:, None], -np.inf)


class BatchedSuppressTokens:
    """openai-whisper's SuppressTokens.apply is `logits[:, <python list>] = -inf`: the list becomes an index tensor (host ->
    device) on every call -- 2.8 ms per decoder call at 256 streams.  Same columns, the index built once."""

    def __init__(self, suppress_tokens):
        self.suppress_tokens = list(suppress_tokens)
        self._idx = None

    def apply(self, logits, tokens):
        if self._idx is None or self._idx.device != logits.device:
            self._idx = torch.tensor(self.suppress_tokens, dtype=torch.long, device=logits.device)
        logits.index_fill_(1, self._idx, -np.inf)


_RULES_PROBED = {}        # backend filter class -> does BatchedTimestampRules reproduce it? (probed once per process)


def _rules_match_backend(rule):
    """The backend's own ApplyTimestampRules and BatchedTimestampRules on a small probe of random token histories (no
    timestamp yet, open and closed pairs, the first sampled position) and logits: the batched form is used only if the
    results are identical -- an older openai-whisper without the 'timestamps never decrease' rule, or a newer one with
    rules this file does not know, keeps its own filter (as the FUSED_ATTENTION / REUSE_DECODER_LOGITS self-checks do)."""
    key = type(rule)
    if key in _RULES_PROBED:
        return _RULES_PROBED[key]
    ok = True
    try:
        tk = rule.tokenizer
        ts0 = int(tk.timestamp_begin)
        V = ts0 + 1501
        rng = np.random.RandomState(1)
        g = torch.Generator().manual_seed(1)
        mine = BatchedTimestampRules.like(rule)
        sb = int(rule.sample_begin)
        for n in (0, 1, 2, 3, 6, 17):
            rows = []
            for _ in range(6):
                seq, t = [], int(rng.randint(0, 60))
                while len(seq) < n:
                    kind = rng.randint(4) if seq else 0
                    if kind == 0:
                        t += int(rng.randint(0, 40))
                        seq.append(ts0 + t)
                    elif kind == 1 and seq[-1] >= ts0:
                        seq.append(seq[-1] if rng.rand() < 0.5 else ts0 + t + int(rng.randint(0, 9)))
                    else:
                        seq.append(int(rng.randint(300, 40000)))
                rows.append([11] * sb + seq)
            tokens = torch.tensor(rows)
            logits = torch.randn((len(rows), V), generator=g) * 3
            logits[:, ts0:] += float(rng.choice([-6.0, 0.0, 6.0]))
            a, b = logits.clone(), logits.clone()
            rule.apply(a, tokens)
            mine.apply(b, tokens)
            if not torch.equal(a, b):
                ok = False
                break
    except Exception as e:                                 # noqa: BLE001 -- a backend this probe cannot drive keeps its own filter
        logger.debug(f"whisper_timestamped: timestamp-rule probe failed ({e!r})")
        ok = False
    if not ok:
        logger.warning("whisper_timestamped: this backend's ApplyTimestampRules differs from the batched form "
                       "(another openai-whisper version?): the B-stream driver keeps the backend's own row-by-row filter")
    _RULES_PROBED[key] = ok
    return ok


def vectorize_filters(task):
    if VECTORIZED_TIMESTAMP_RULES:
        task.logit_filters = [BatchedTimestampRules.like(f) if type(f).__name__ == "ApplyTimestampRules" and _rules_match_backend(f)
                              else BatchedSuppressTokens(f.suppress_tokens) if type(f).__name__ == "SuppressTokens" and hasattr(f, "suppress_tokens")
                              else f for f in task.logit_filters]
    return task


class _RowAlreadyInTheRing:
    """What a stream's session is handed as `outs` for a replayed decoder call other than the prompt call: it only ever
    takes `outs[0, -1]` of it, to append a row that the driver has already written into the stream's block."""

    def __getitem__(self, key):
        return None


_IN_RING = _RowAlreadyInTheRing()


# ----------------------------------------------------------------------------------------------------------------------
# recorder: the hooks of ONE batched decoder loop
# ----------------------------------------------------------------------------------------------------------------------
class _Recorder:
    """Forward hooks for the duration of one ``_main_loop`` over g streams.  Per decoder call it keeps the input token
    ids (host), writes the QK rows of all streams (one launch) and, one call later -- when the sampler has filtered the
    previous call's last-position logits in place and the tokens it sampled are this call's input -- takes the digest
    of those (g, V) rows (one launch: StreamRings.write_digest)."""

    def __init__(self, model, rings: StreamRings, hooked_blocks, ring_index, verify: bool):
        self.model, self.rings, self.hooked_blocks = model, rings, hooked_blocks
        self.ring_index = ring_index                       # device int32[g]
        self.ring_index_long = ring_index.long()
        self.calls = []                                    # per decoder call: list of g token lists
        self.pending = None
        self.capture = True                                # False during language detection (T.py:828: not started)
        self.lang_outs = None
        self.verify = verify
        self.verify_rows = None
        self.verified = None
        self._q = [None] * len(hooked_blocks)
        self._k = [None] * len(hooked_blocks)
        self._v = [None] * len(hooked_blocks)
        self.fused_checked = False
        self.hooks = []

    def install(self):
        m = self.model
        self.hooks.append(m.decoder.token_embedding.register_forward_hook(self.on_tokens))
        used = self.rings.used
        for j in used:
            ca = m.decoder.blocks[self.hooked_blocks[j]].cross_attn
            self.hooks.append(ca.query.register_forward_hook(lambda mod, i, o, index=j: self._q.__setitem__(index, o)))
            self.hooks.append(ca.key.register_forward_hook(lambda mod, i, o, index=j: self._k.__setitem__(index, o)))
            self.hooks.append(ca.value.register_forward_hook(lambda mod, i, o, index=j: self._v.__setitem__(index, o)))
        if used:
            ca = m.decoder.blocks[self.hooked_blocks[used[-1]]].cross_attn
            self.hooks.append(ca.register_forward_hook(self.on_last_cross_attention))
        self.hooks.append(m.decoder.ln.register_forward_hook(self.on_ln))
        self.hooks.append(m.decoder.register_forward_hook(self.on_logits))
        return self

    def remove(self):
        for h in self.hooks:
            h.remove()
        self.hooks = []

    # --- hooks, in firing order within one decoder call
    def on_tokens(self, layer, ins, outs):
        self.commit(ins[0][:, -1])                         # this call's input = what the sampler drew from the pending rows
        self.calls.append(ins[0].tolist())                 # the per-step host read whisper's own loop needs anyway

    def on_last_cross_attention(self, layer, ins, outs):
        if not self.capture:
            return
        row = len(self.calls) - 1
        self.rings.write_qk(self._q, self._k, self.ring_index, row)
        if not self.fused_checked:
            self.check_fused_rows(row)

    def on_ln(self, layer, ins, outs):
        """First call of the loop: the reference-way rows (T.py:871-874: a second projection) kept aside, to be compared
        with what the sampler leaves in the decoder's own logits (REUSE_DECODER_LOGITS = "auto", efficient.py)."""
        if self.verify and self.capture and len(self.calls) == 1:
            e = torch.transpose(self.model.decoder.token_embedding.weight, 0, 1).to(outs.dtype)
            self.verify_rows = (outs[:, -1, :] @ e).float()

    def on_logits(self, layer, ins, outs):
        if not self.capture:
            self.lang_outs = outs                          # language detection: (n, 1, V), read by every new stream's session
        self.pending = outs

    def commit(self, sampled, final=False):
        """The previous call's last-position rows are final (the sampler filtered them in place) and `sampled` (g,) holds
        the tokens drawn from them: their digests into the ring.  The rows themselves are not kept -- except those of the
        loop's LAST call (`final`), one row per stream."""
        if self.pending is None or not self.capture:
            self.pending = None
            return
        rows = self.pending[:, -1]
        step = len(self.calls) - 1
        if self.verify and step == 0 and self.verify_rows is not None:
            self.verified = (rows.clone(), self.verify_rows)       # compared by the driver with the filters applied
        self.rings.write_digest(rows, sampled, self.ring_index, step)
        if final:
            self.rings.keep_last_rows(rows, self.ring_index_long)
        self.pending = None

    def check_fused_rows(self, row):
        """Once per run (as EfficientSession._check_fused_rows): the rows wt_qk_rows_streams computes from
        cross_attn.query / .key against the backend's own unfused attention, for the first stream of the call."""
        self.fused_checked = True
        worst = 0.0
        with torch.no_grad(), backend.attention_weights_exposed(True):
            for index in self.rings.used:
                if self._v[index] is None:
                    continue
                ca = self.model.decoder.blocks[self.hooked_blocks[index]].cross_attn
                out = ca.qkv_attention(self._q[index][:1, -1:], self._k[index][:1], self._v[index][:1])
                qk = out[1] if isinstance(out, tuple) and len(out) > 1 else None
                if qk is None:
                    continue
                sl, sh, ss = self.rings.sel
                mine = (sl == self.rings.used.index(index))
                want = qk[0, sh[mine].long(), -1].float()
                got = self.rings.qk[int(self.ring_index[0]), ss[mine].long(), row].float()
                worst = max(worst, float((got - want).abs().max()))
        self._v = [None] * len(self.hooked_blocks)
        q_any = next(q for q in self._q if q is not None)
        tol = 2e-3 if (q_any.dtype == torch.float32 and self.rings.qk.dtype == torch.float32) else 0.1
        if not worst <= tol:
            raise RuntimeError(f"FUSED_ATTENTION self-check failed: the QK rows computed from cross_attn.query/key differ "
                               f"from the backend's own unfused attention by {worst:.3g} (> {tol})")


# ----------------------------------------------------------------------------------------------------------------------
# one recording
# ----------------------------------------------------------------------------------------------------------------------
class _Stream:
    """openai-whisper's transcribe() loop state of one recording (whisper/transcribe.py: seek, prompt, segments)."""

    def __init__(self, index, mel, content_frames, session, tokenizer, language, initial_prompt_tokens):
        self.index, self.mel, self.content_frames = index, mel, content_frames
        self.session, self.tokenizer, self.language = session, tokenizer, language
        self.seek = 0
        self.all_tokens = list(initial_prompt_tokens)
        self.n_initial_prompt = len(initial_prompt_tokens)
        self.all_segments = []
        self.prompt_reset_since = 0
        self.done = False
        self.block = index                  # its block of the shared rings (the driver assigns it on admission)
        # the window being decoded
        self.segment_size = 0
        self.task = None
        self.initial_tokens = None

    def active(self):
        return not self.done and self.content_frames > 0 and self.seek < self.content_frames

    def window_mel(self):
        self.segment_size = min(N_FRAMES, self.content_frames - self.seek)
        seg = self.mel[:, self.seek:self.seek + self.segment_size]
        if seg.shape[-1] < N_FRAMES:
            seg = torch.nn.functional.pad(seg, (0, N_FRAMES - seg.shape[-1]))
        return seg

    def take_result(self, tokens, avg_logprob, no_speech_prob, temperature, opts, w):
        """What whisper's loop does with the DecodingResult of a window: no-speech skip, segments at consecutive
        timestamps, seek, prompt bookkeeping."""
        tk = self.tokenizer
        input_stride = N_FRAMES // 1500
        time_precision = input_stride * HOP_LENGTH / SAMPLE_RATE
        time_offset = float(self.seek * HOP_LENGTH / SAMPLE_RATE)
        segment_duration = self.segment_size * HOP_LENGTH / SAMPLE_RATE
        text = tk.decode(tokens).strip()
        result = dict(temperature=temperature, avg_logprob=avg_logprob, no_speech_prob=no_speech_prob,
                      compression_ratio=w.utils.compression_ratio(text))
        nst, lpt = opts["no_speech_threshold"], opts["logprob_threshold"]
        if nst is not None:
            should_skip = no_speech_prob > nst
            if lpt is not None and avg_logprob > lpt:
                should_skip = False
            if should_skip:
                self.seek += self.segment_size
                return
        seek0 = self.seek

        def new_segment(start, end, toks):
            text_tokens = [t for t in toks if t < tk.eot]
            return {"seek": seek0, "start": start, "end": end, "text": tk.decode(text_tokens), "tokens": list(toks), **result}

        toks = list(tokens)
        is_ts = [t >= tk.timestamp_begin for t in toks]
        single_timestamp_ending = is_ts[-2:] == [False, True]
        consecutive = [i + 1 for i in range(len(toks) - 1) if is_ts[i] and is_ts[i + 1]]
        current = []
        if consecutive:
            slices = list(consecutive)
            if single_timestamp_ending:
                slices.append(len(toks))
            last = 0
            for cur in slices:
                sl = toks[last:cur]
                t0, t1 = sl[0] - tk.timestamp_begin, sl[-1] - tk.timestamp_begin
                current.append(new_segment(time_offset + t0 * time_precision, time_offset + t1 * time_precision, sl))
                last = cur
            if single_timestamp_ending:
                self.seek += self.segment_size
            else:
                self.seek += (toks[last - 1] - tk.timestamp_begin) * input_stride
        else:
            duration = segment_duration
            stamps = [t for t, f in zip(toks, is_ts) if f]
            if stamps and stamps[-1] != tk.timestamp_begin:
                duration = (stamps[-1] - tk.timestamp_begin) * time_precision
            current.append(new_segment(time_offset, time_offset + duration, toks))
            self.seek += self.segment_size
        for seg in current:
            if seg["start"] == seg["end"] or seg["text"].strip() == "":
                seg["text"], seg["tokens"], seg["words"] = "", [], []
        self.all_segments.extend({"id": i, **seg} for i, seg in enumerate(current, start=len(self.all_segments)))
        self.all_tokens.extend(t for seg in current for t in seg["tokens"])
        if not opts["condition_on_previous_text"] or temperature > 0.5:
            self.prompt_reset_since = len(self.all_tokens)

    def transcription(self):
        return dict(text=self.tokenizer.decode(self.all_tokens[self.n_initial_prompt:]), segments=self.all_segments,
                    language=self.language)


# ----------------------------------------------------------------------------------------------------------------------
# the driver
# ----------------------------------------------------------------------------------------------------------------------
def supports(whisper_options, vad=None, naive_approach=False, plot_word_alignment=False):
    """Can this call take the B-stream path?  (Else: one transcribe_timestamped() per recording.)"""
    t = whisper_options.get("temperature", 0.0)
    return not (naive_approach or vad or plot_word_alignment or isinstance(t, (list, tuple)) or
                whisper_options.get("beam_size") is not None or (whisper_options.get("best_of") or 1) > 1)


def backend_missing():
    """What the B-stream driver needs of the ASR backend beyond what the one-stream path uses: openai-whisper's
    DecodingTask with its decoder loop as a method of its own (whisper/decoding.py: `_main_loop(audio_features, tokens)`,
    `_get_audio_features(mel)`, `initial_tokens`, `sample_begin`, `decoder.finalize`, `logit_filters`).  -> list of what is
    missing ([] = fine); transcribe_batch then decodes the recordings one stream at a time instead."""
    w = backend.whisper()
    task = getattr(getattr(w, "decoding", None), "DecodingTask", None)
    if task is None:
        return ["whisper.decoding.DecodingTask"]
    missing = [f"DecodingTask.{m}" for m in ("_main_loop", "_get_audio_features", "_get_initial_tokens") if not hasattr(task, m)]
    if not hasattr(w, "DecodingOptions"):
        missing.append("whisper.DecodingOptions")
    if not hasattr(getattr(w, "utils", None), "compression_ratio"):
        missing.append("whisper.utils.compression_ratio")
    return missing


def _length_hint(audio):
    """Samples of a recording given as an array / tensor; for a path, its file size (files of one codec scale with their
    duration: good enough to order them)."""
    if isinstance(audio, str):
        import os
        try:
            return os.path.getsize(audio)
        except OSError:
            return 0
    return int(audio.shape[-1]) if hasattr(audio, "shape") and len(audio.shape) else 0


def _whole_file_mels(model, audios, dtype):
    """Every recording's log-mel over the whole file + 30 s of silence, file-global max clamp (whisper's transcribe():
    log_mel_spectrogram(audio, n_mels, padding=N_SAMPLES)) on the GPU; recordings of equal length share a launch."""
    import sys
    from . import audio as wt_audio
    from .naive import get_audio_tensor
    dev = model.device
    pcm = [get_audio_tensor(a).float().reshape(-1) for a in audios]
    mels = [None] * len(pcm)
    # the same choice backend.gpu_log_mel makes for the one-stream path: the HIP front end when it reproduces this
    # backend's log_mel_spectrogram on a probe signal, else the backend's own function
    theirs = getattr(sys.modules.get("whisper.transcribe"), "log_mel_spectrogram", None) or backend.whisper().log_mel_spectrogram
    if not (efficient.GPU_FRONT_END and backend._front_end_matches(theirs, wt_audio, dev, model.dims.n_mels)):
        return [theirs(p, model.dims.n_mels, padding=N_SAMPLES).to(dev).to(dtype) for p in pcm]
    by_len = {}
    for i, p in enumerate(pcm):
        by_len.setdefault(int(p.numel()), []).append(i)
    fb = wt_audio.mel_filters(dev, model.dims.n_mels)
    for n, idx in by_len.items():
        if n == 0:
            for i in idx:
                mels[i] = wt_audio.log_mel_spectrogram(pcm[i], n_mels=model.dims.n_mels, padding=N_SAMPLES, device=dev)
            continue
        batch = torch.zeros((len(idx), n + N_SAMPLES), dtype=torch.float32, device=dev)
        for j, i in enumerate(idx):
            batch[j, :n] = pcm[i].to(dev, non_blocking=True)
        mel, _ = _lib.logmel(batch, fb, None, n_frames=(n + N_SAMPLES) // HOP_LENGTH)
        for j, i in enumerate(idx):
            mels[i] = mel[j]
    return [m.to(dtype) for m in mels]


def transcribe_efficient_streams(model, audios, *, remove_punctuation_from_words, compute_word_confidence,
                                 include_punctuation_in_confidence, refine_whisper_precision_nframes, alignment_heads,
                                 plot_word_alignment, word_alignment_most_top_layers, detect_disfluencies,
                                 trust_whisper_timestamps, use_timestamps_for_alignment=True, max_streams=32,
                                 **whisper_options):
    """``[transcribe_efficient(model, a, ...) for a in audios]`` with up to ``max_streams`` recordings stepping through
    the decoder together; a recording that is finished hands its place to the next one (continuous admission: short and
    long recordings mix without the long ones ending up alone).  Returns a list of (transcription, words), one per
    recording, in order."""
    assert supports(whisper_options, plot_word_alignment=plot_word_alignment)
    return _run_streams(model, list(audios), dict(whisper_options), max(1, int(max_streams)),
                        remove_punctuation_from_words=remove_punctuation_from_words,
                        compute_word_confidence=compute_word_confidence,
                        include_punctuation_in_confidence=include_punctuation_in_confidence,
                        refine_whisper_precision_nframes=refine_whisper_precision_nframes,
                        alignment_heads=alignment_heads, word_alignment_most_top_layers=word_alignment_most_top_layers,
                        detect_disfluencies=detect_disfluencies, trust_whisper_timestamps=trust_whisper_timestamps,
                        use_timestamps_for_alignment=use_timestamps_for_alignment)


def _run_streams(model, audios, opts, max_streams, **session_kwargs):
    w = backend.whisper()
    dev = model.device
    _lib.require_gpu(dev)
    N = len(audios)
    S = min(N, max_streams)                               # ring blocks = recordings in flight
    opts["verbose"] = None
    fp16 = bool(opts.get("fp16"))
    dtype = torch.float16 if fp16 else torch.float32
    decode_keys = ("task", "language", "sample_len", "best_of", "beam_size", "patience", "length_penalty", "suppress_tokens",
                   "fp16")                               # what whisper's transcribe() hands to DecodingOptions
    temperature = opts["temperature"]

    n_blocks = len(model.decoder.blocks)
    top_layers = session_kwargs["word_alignment_most_top_layers"]
    top = n_blocks if top_layers is None else min(top_layers, n_blocks)
    hooked_blocks = list(range(n_blocks - top, n_blocks))
    tk0 = backend.get_tokenizer(model, task=opts["task"], language=opts["language"] or "en")     # (special-token ids: the same for every language)
    per_stream = StreamRings.bytes_per_stream(model, session_kwargs["alignment_heads"], hooked_blocks, efficient.RING_DTYPE,
                                              opts.get("sample_len"), tk0)
    if dev.type == "cuda":
        # the rings must fit: S streams x (QK rows + digests) next to the model, its KV caches and the whole-file log-mels
        # (free = what the driver reports + what torch's caching allocator holds without using it: a second batch in
        #  the same process finds the previous rings there)
        free_bytes = torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
        fit = int(0.5 * free_bytes // max(per_stream, 1))
        if fit < S:
            logger.warning(f"whisper_timestamped: {S} decoder streams need {S * per_stream / 2**30:.1f} GiB of ring memory; "
                           f"{max(fit, 1)} fit the free device memory -- the other recordings are admitted as streams finish")
            S = max(1, fit)
    rings = StreamRings(model, session_kwargs["alignment_heads"], hooked_blocks, S, efficient.RING_DTYPE,
                        sample_len=opts.get("sample_len"), tokenizer=tk0)
    sink = _Sink(default_workspace(dev))
    streams = [None] * N                                  # by recording
    free = list(range(S))
    # longest recordings first (as sharding.partition_units deals units to ranks): a long recording admitted last would
    # decode alone at the end.  Results go back in the caller's order whatever the order of admission.
    pending = sorted(range(N), key=lambda i: (-_length_hint(audios[i]), i)) if N > S else list(range(N))
    admissions = 0

    def admit():
        """Pending recordings into the free ring blocks: their whole-file log-mels (one launch per length), their
        language (one batched detect_language call), their sessions on views of their blocks."""
        nonlocal admissions
        new = []
        while free and pending:
            new.append((pending.pop(0), free.pop(0)))
        if not new:
            return
        admissions += 1
        mels = _whole_file_mels(model, [audios[i] for i, _ in new], dtype)
        languages = [opts["language"]] * len(new)
        lang_events = None
        if opts["language"] is None:                      # whisper's transcribe(): detect on the first 30 s when not given
            if not model.is_multilingual:
                languages = ["en"] * len(new)
            else:
                first = torch.stack([torch.nn.functional.pad(m[:, :N_FRAMES], (0, max(0, N_FRAMES - m.shape[-1]))) for m in mels])
                rec = _Recorder(model, rings, hooked_blocks, torch.tensor([b for _, b in new], dtype=torch.int32, device=dev),
                                verify=False)
                rec.capture = False
                rec.install()
                try:
                    _, probs = model.detect_language(first)
                finally:
                    rec.remove()
                languages = [max(p, key=p.get) for p in probs]
                lang_events = (first, rec.calls[0], rec.lang_outs, _lib.HostCopy(_lib.find_start_padding(first.float())))
        for j, ((i, block), mel, lang) in enumerate(zip(new, mels, languages)):
            session = EfficientSession(model, dict(opts), ring=_QKView(rings.qk[block]), logits=_LogitsView(rings, block),
                                       sink=sink, **session_kwargs)
            tk = backend.get_tokenizer(model, task=opts["task"], language=lang)
            prompt0 = tk.encode(" " + opts["initial_prompt"].strip()) if opts.get("initial_prompt") is not None else []
            st = streams[i] = _Stream(i, mel, mel.shape[-1] - N_FRAMES, session, tk, lang, prompt0)
            st.block = block
            if lang_events is not None:                   # the language-detection call, as this stream's hooks saw it
                first, calls, outs, pad0 = lang_events
                session.hook_mel(None, (first[j:j + 1],), None, pad_handle=_SliceOfCopy(pad0, j))
                session.on_tokens(list(calls[j]))
                session.hook_decoder_logits(None, None, outs[j:j + 1])

    def retire(st):
        """The backend would return now: the stream's last window is flushed and closed (its units wait in the sink for
        the next launch set, which is queued BEFORE anything can overwrite its ring block), the block is free."""
        st.session.end_of_stream()
        st.done = True
        st.mel = None
        free.append(st.block)

    rounds = groups = 0
    loop_sizes = []
    verify = efficient.REUSE_DECODER_LOGITS == "auto"
    fused_checked = False
    # The driver's loops allocate a few thousand small ACYCLIC objects per stream and window (token lists, word dictionaries,
    # views): reference counting frees them; the cyclic collector only re-scans the process's long-lived objects (the model's
    # modules, torch's own tables) every few thousand allocations -- a tenth of a 256-stream loop's host time
    # (profiles/r5d_streams_gc.txt).  Paused for the duration of the batch, restored (and run once) at its end.
    try:
        with torch.no_grad(), paused_gc() as gc_pause:
            while True:
                admit()
                for st in streams:                            # (recordings with nothing to decode: empty audio)
                    if st is not None and not st.done and not st.active():
                        retire(st)
                act = [st for st in streams if st is not None and st.active()]
                if not act:
                    if pending:
                        continue
                    break
                rounds += 1
                gc_pause.tick(rounds)
                # ---- every active stream's prompt for its next window (the backend's own DecodingTask builds it)
                for st in act:
                    kwargs = {k: opts[k] for k in decode_keys if k in opts}
                    kwargs["language"] = st.language
                    kwargs["prompt"] = st.all_tokens[st.prompt_reset_since:]
                    if temperature > 0:
                        kwargs.pop("beam_size", None), kwargs.pop("patience", None)
                    else:
                        kwargs.pop("best_of", None)
                    st.task = w.decoding.DecodingTask(model, w.DecodingOptions(**kwargs, temperature=temperature))
                    st.initial_tokens = list(st.task.initial_tokens)
                # ---- this round's windows, their padding, every stream's prompt call (closes the previous window)
                mel_batch = torch.stack([st.window_mel() for st in act]).to(dtype)
                pad = _lib.HostCopy(_lib.find_start_padding(mel_batch.float()))
                for j, st in enumerate(act):
                    st.session.hook_mel(None, (mel_batch[j:j + 1],), None, pad_handle=_SliceOfCopy(pad, j))
                    st.session.on_tokens(list(st.initial_tokens))
                sink.launch()
                # ---- one batched decoder loop per initial-token length
                by_len = {}
                for j, st in enumerate(act):
                    by_len.setdefault(len(st.initial_tokens), []).append(j)
                for L, members in by_len.items():
                    groups += 1
                    loop_sizes.append(len(members))
                    grp = [act[j] for j in members]
                    if ON_GROUP_DECODE is not None:
                        ON_GROUP_DECODE([st.index for st in grp])
                    task = vectorize_filters(grp[0].task)
                    ring_index = torch.tensor([st.block for st in grp], dtype=torch.int32, device=dev)
                    rec = _Recorder(model, rings, hooked_blocks, ring_index, verify=verify)
                    rec.fused_checked = fused_checked or not efficient.FUSED_ATTENTION
                    task.decoder.reset()
                    rec.install()
                    try:
                        feats = task._get_audio_features(mel_batch[members])
                        tokens0 = torch.tensor([st.initial_tokens for st in grp], device=dev)
                        tokens, sum_logprobs, no_speech = task._main_loop(feats, tokens0)
                        rec.commit(tokens[:, -1], final=True)  # the last call's rows and what was sampled from them
                    finally:
                        rec.remove()
                    fused_checked = True
                    _finish_group(grp, task, rec, rings, tokens, sum_logprobs, no_speech, temperature, opts, w, verify)
                for st in act:                                # finished recordings hand their blocks to the next ones
                    if not st.active():
                        retire(st)
            sink.resolve()
            out = [st.session.compiled(st.transcription()) for st in streams]
    finally:
        for batch, _ in sink.in_flight:           # (an error mid-run: the launched batches hand their buffers back)
            batch.release()
        sink.in_flight = []
    LAST_RUN.clear()
    LAST_RUN.update(streams=N, ring_blocks=S, admissions=admissions, rounds=rounds, decoder_loops=groups,
                    alignment_launch_sets=sink.launch_sets, streams_per_loop=loop_sizes,
                    ring_bytes=int(rings.qk.numel() * rings.qk.element_size() + 4 * (rings.digest.numel() + rings.slice.numel())))
    return out


def _finish_group(grp, task, rec, rings, tokens, sum_logprobs, no_speech, temperature, opts, w, verify):
    """After a batched decoder loop: results per stream, every stream's digest records of the loop in ONE host copy
    (the chosen-token log-probabilities among them: computed call by call while the loop ran), the loop's recorded
    calls replayed into every stream's session, the backend's window bookkeeping."""
    tk = task.tokenizer
    g = len(grp)
    L = task.sample_begin
    n_calls = len(rec.calls)
    # one copy for the group: (g, n_calls, 8) digest records -> page-locked host memory
    digests = _lib.HostCopy(rings.digest[rec.ring_index_long, :n_calls])
    tokens_f, sums = task.decoder.finalize(tokens.reshape(g, 1, -1), sum_logprobs.reshape(g, 1))
    rows_host = tokens.tolist()
    # each stream takes part in the calls up to the one whose sample was <|endoftext|> (later calls feed it eot)
    sampled, results = [], []
    for i, s in enumerate(grp):
        t = tokens_f[i][0]
        t = t.tolist() if hasattr(t, "tolist") else list(t)
        body = t[L:]
        cut = body.index(tk.eot) if tk.eot in body else len(body)
        out_tokens = body[:cut]
        lp = sums[i][0] if isinstance(sums[i], (list, tuple)) else sums[i]
        results.append((out_tokens, float(lp) / (len(out_tokens) + 1)))
        smp = rows_host[i][L:]                               # what the sampler returned at calls 0, 1, ...
        n_mine = min(n_calls, (smp.index(tk.eot) + 1) if tk.eot in smp else len(smp))
        sampled.append(smp[:n_mine])
    host = digests.wait().numpy().reshape(g, n_calls, _lib.DIGEST_WORDS)
    # the sampler's in-place filtering must be there (<|notimestamps|> is always suppressed) -- read from the digests
    if tk.no_timestamps is not None:
        k = 4 + rings.aux_tokens.index(int(tk.no_timestamps))
        if not bool(np.isneginf(host[:, 0, k]).all()):
            raise RuntimeError("streams: this backend does not filter the decoder's logits in place; use the one-stream path "
                               "(whisper_timestamped.efficient.REUSE_DECODER_LOGITS = False)")
    if verify and rec.verified is not None:
        # "auto": the first token of the window both ways, for every stream (same -inf pattern, same values)
        got, want = rec.verified
        context = torch.tensor([s.initial_tokens for s in grp], device=got.device)
        want = want.clone()
        for f in task.logit_filters:
            f.apply(want, context)
        fin = torch.isfinite(want)
        tol = 1e-3 if want.dtype == torch.float32 and not opts.get("fp16") else 5e-2
        if not (bool(torch.equal(fin, torch.isfinite(got))) and bool(((got[fin] - want[fin]).abs() <= tol).all())):
            raise RuntimeError("streams: the decoder's logits do not carry the sampler's filtering (or differ from the "
                               "re-projected ones); use the one-stream path")
    # replay: call 0 was the prompt (its token half ran before the loop); then the remaining calls of each stream
    no_speech = [float(x) for x in no_speech]
    for i, s in enumerate(grp):
        ses = s.session
        assert rec.calls[0][i] == s.initial_tokens
        n = len(sampled[i])
        ses.logits.window(host[i, :n], sampled[i], full_row=(n_calls - 1) if n == n_calls else None)
        ses.replay_prompt_logits(no_speech[i])
        mine = sampled[i][:n - 1]                             # the one token fed at calls 1, 2, ...: what call k - 1 sampled
        ts0, k, n = ses.tokenizer.timestamp_begin, 0, len(mine)
        while k < n:
            e = k
            while e < n and mine[e] < ts0:
                e += 1
            if e - k >= 2 and ses.text_run_is_plain(e - k):         # a run of text tokens: booked at once
                ses.on_text_run(mine[k:e], _IN_RING)
                k = e
                continue
            ses.on_tokens([mine[k]])
            ses.hook_decoder_logits(None, None, _IN_RING)
            k += 1
        out_tokens, avg_logprob = results[i]
        s.take_result(out_tokens, avg_logprob, no_speech[i], temperature, opts, w)
