"""Efficient strategy: word alignment on the fly, while openai-whisper decodes.

Behavioural mirror of ``_transcribe_timestamped_efficient``
(/root/reference/whisper_timestamped/transcribe.py:359-1001; the hook state
machine is described in SURVEY.md Appendix B).  Control flow (when a segment is
flushed, the fallbacks for a stuck decoder, the retroactive no-speech skip)
stays host Python and follows the reference decision by decision; the DATA
plane is what changed for the MI355X:

* cross-attention rows go into a device ring (``QKCaptureRing``), alignment-head
  rows only, instead of a host copy of every head per token (:793);
* filtered logits go into a device ring (``LogitsRing``); log-softmax + gather of
  the chosen tokens is one kernel per 30 s window instead of a stored (V,)
  vector per step (:875-876) and a Python loop of 0-d tensor indexing (:735);
* a flushed segment is aligned by the HIP cost + DTW kernels straight from the
  ring (strided view, no concatenation of per-step tensors, :516-525).
"""
from __future__ import annotations

import logging
import sys

import numpy as np
import torch

from . import _lib, backend
from .alignment import AlignmentBatch, default_workspace, head_pairs, planned_words, prepare_unit, set_padding
from .capture import LogitsRing, QKCaptureRing
from .confidence import segment_confidences
from .words import HOP_LENGTH, N_AUDIO_CTX, SAMPLE_RATE

logger = logging.getLogger("whisper_timestamped")

# what the last EfficientSession did (tests / diagnostics): windows whose first token verified the logits reuse,
# whether the session fell back to the reference's per-token projection, alignment launch sets
LAST_SESSION = {}

# storage type of the captured cross-attention rows: float32 = what the reference sees (qk.float());
# float16 halves the HBM bytes of the cost kernel (build-side option, BASELINE config 5; parity quantified in tests)
RING_DTYPE = torch.float32
# compute whisper's whole-file log-mel with the HIP front end instead of torch.stft (backend.gpu_log_mel)
GPU_FRONT_END = True
# Take the step's filtered logits from the decoder itself instead of re-projecting ln's output and re-applying the logit
# filters as the reference does (transcribe.py:871-874 duplicates, once per token, a D x V GEMV and the filters
# whisper's sampler has just applied IN PLACE to the tensor the decoder returned).  The row is read one step later, when
# the sampler is done with it.  Same values up to GEMM-vs-GEMV rounding of the first step of a window.
#   "auto" (default): the first token of every 30 s window is done BOTH ways and the two rows are compared (same -inf
#       pattern, same values); if they agree the rest of the window reuses the decoder's rows, if they ever do not
#       (a backend that does not filter in place, other filters ...) the session goes back to the reference's way
#       for good -- nothing is lost, the verified row of that token is the reference-way one;
#   True: always reuse (raises if the backend visibly does not filter in place);  False: the reference's way.
REUSE_DECODER_LOGITS = "auto"
# Compute the alignment heads' QK rows from cross_attn.query / cross_attn.key outputs (wt_qk_rows) and let the backend
# keep its fused attention.  The reference reads qk from MultiHeadAttention's second output, which only exists on the
# unfused path: it runs EVERY attention module of the model (encoder included) unfused, inside
# whisper.model.disable_sdpa().  False = exactly that (the reference's attention arithmetic for the whole model).
FUSED_ATTENTION = True
# Align the segments of a 30 s window with ONE launch set when the window closes, and read the (KB-sized) result
# record one window later, instead of one launch set + one device->host wait per segment: the words' texts and
# tokens are known from the token split alone, only their times are filled in later.  Off (or detect_disfluencies,
# whose "[*]" words only exist after the kernels ran) = the reference's shape, a synchronous alignment per segment.
DEFER_ALIGNMENT = True


class _RowInTheRing:
    """`outs` of a replayed decoder call (streams.py): the session only ever takes `outs[0, -1]` of it, to append a row
    whose digest the driver has already written into the stream's block."""

    def __getitem__(self, key):
        return None


_ROW_IN_THE_RING = _RowInTheRing()


class EfficientSession:
    def __init__(self, model, whisper_options, *, remove_punctuation_from_words, compute_word_confidence,
                 include_punctuation_in_confidence, refine_whisper_precision_nframes, alignment_heads,
                 word_alignment_most_top_layers, detect_disfluencies, trust_whisper_timestamps,
                 use_timestamps_for_alignment=True, ring_dtype=None, ring=None, logits=None, sink=None, plot=False):
        """``ring`` / ``logits`` / ``sink``: the B-stream form (streams.py).  One session per decoder stream, fed with the
        RECORDED events of a batched decoder call instead of live hooks: its attention rows and logits rows live in its
        block of rings shared by all streams (views handed in here), and the alignment units of all streams of a window
        set go out in ONE launch set (the sink) instead of one per stream."""
        self.model = model
        self.opts = whisper_options
        self.remove_punctuation_from_words = remove_punctuation_from_words
        self.compute_word_confidence = compute_word_confidence
        self.include_punctuation_in_confidence = include_punctuation_in_confidence
        self.refine_nframes = refine_whisper_precision_nframes
        self.alignment_heads = alignment_heads
        self.detect_disfluencies = detect_disfluencies
        self.trust = trust_whisper_timestamps
        self.use_timestamps_for_alignment = use_timestamps_for_alignment
        self.plot = plot                 # plot_word_alignment: a figure per aligned segment (plotting.py)

        self.temperature = whisper_options["temperature"]
        self.no_speech_threshold = whisper_options["no_speech_threshold"]
        self.logprob_threshold = whisper_options["logprob_threshold"]
        self.language = whisper_options["language"]
        self.tokenizer = backend.get_tokenizer(model, task=whisper_options["task"], language=self.language)
        self.logit_filters = backend.get_logit_filters(model, whisper_options)
        self.max_sample_len = whisper_options["sample_len"] or model.dims.n_text_ctx // 2
        self.n_ctx = model.dims.n_text_ctx
        self.new_whisper = backend.ge_20230306()

        n_blocks = len(model.decoder.blocks)
        top = n_blocks if word_alignment_most_top_layers is None else min(word_alignment_most_top_layers, n_blocks)
        self.hooked_blocks = list(range(n_blocks - top, n_blocks))
        dev = model.device
        _lib.require_gpu(dev)
        self.replay = ring is not None
        self.sink = sink
        self.ring = ring if ring is not None else \
            QKCaptureRing(dev, head_pairs(alignment_heads), len(self.hooked_blocks), model.dims.n_text_head,
                          n_ctx=model.dims.n_audio_ctx, capacity=self.n_ctx, dtype=ring_dtype or RING_DTYPE)
        self.logits = logits if logits is not None else LogitsRing(dev, model.dims.n_vocab, capacity=self.n_ctx + 1)
        self.embedding_t = None

        # outcome
        self.words_per_segment = []      # timestamped_word_segments
        self.segment_tokens = [[]]       # last entry = the segment being decoded
        self.segment_avglogprobs = []
        self.segment_logprobs = []
        self.language_probs = None
        # open-segment / window state
        self.open_rows = []              # ring rows of the open segment (one per token of segment_tokens[-1])
        self.row_next = 0
        self.sot_index = None
        self.no_speech_prob = None
        self.window_inputs = []          # chunk_tokens: every decoder input of the window (lists of ids)
        self.window_tokens_nosot = []    # chunk_tokens_nosot
        self.last_chunk_token = None
        self.last_token_fallback = None
        self.has_started = False
        self.mfcc = None
        self.new_mfcc = None
        self.saw_consecutive_timestamps = False
        self.first_segment_of_window = 0  # index_begin_30sec_chunck
        self._pad_cache = (None, None)
        self.detected_language = False
        self.pending_logits = None
        self._q = [None] * len(self.hooked_blocks)
        self._k = [None] * len(self.hooked_blocks)
        self._v = [None] * len(self.hooked_blocks)     # only until the fused path has been checked once
        self._fused_checked = False
        self._value_hooks = []
        self.reuse = REUSE_DECODER_LOGITS
        self.reuse_state = "verify"          # "auto": verify (first token of a window) -> trusted | off (for good)
        self._verify_row = None
        self.stats = dict(reuse_mode=self.reuse, windows_verified=0, reuse_fell_back=False, alignment_launch_sets=0)
        # deferred alignment: units queued during the window, launched when it closes, collected one window later
        self.defer = DEFER_ALIGNMENT and not detect_disfluencies
        self.workspace = default_workspace(dev)    # pinned staging + result buffers survive from call to call
        self.queued = []                 # [(unit, placeholder words, padding handle)] of the window being decoded
        self.in_flight = []              # [(AlignmentBatch, [placeholder word lists])] launched, not yet read
        self._pad_handles = {}           # id(mfcc) -> (mfcc, HostCopy of find_start_padding): queued when the mel appears
        # every decoder input of the window, device resident (the logit filters' `tokens` argument)
        self.ctx_buf = None if self.replay else torch.zeros((1, 2 * self.n_ctx + 8), dtype=torch.int64, device=dev)
        self.ctx_len = 0
        if self.replay:                      # the driver has verified the batch's rows; a stream only reads them
            self.reuse = True

    # ------------------------------------------------------------------ small predicates
    def _is_sot(self, cur):
        return cur is None or len(cur) > 1 or cur[0] == self.tokenizer.sot

    def _reached_decoding_limit(self):
        n = len(self.window_tokens_nosot) + 1
        m = n + (len(self.window_inputs[0]) if self.window_inputs else 0)
        return n + 1 >= self.max_sample_len or m > self.n_ctx

    def _is_ts(self, tok):
        return tok >= self.tokenizer.timestamp_begin

    # ------------------------------------------------------------------ hooks
    def hook_mel(self, layer, ins, outs, pad_handle=None):
        """``pad_handle`` (streams.py): an object whose ``wait()[0]`` is find_start_padding of this window, computed for
        all streams of a round in one launch; without it the detector is queued here for this window alone."""
        self.new_mfcc = ins[0]
        if self.mfcc is None:
            self.mfcc = self.new_mfcc
        if self.defer:                    # where this window's zero padding starts: on the host long before it is needed
            self._pad_handles = {k: v for k, v in self._pad_handles.items() if v[0] is self.mfcc}
            if pad_handle is not None:
                self._pad_handles[id(self.new_mfcc)] = (self.new_mfcc, pad_handle)
            else:
                self._padding_handle(self.new_mfcc)

    def _padding_handle(self, mfcc):
        """(mfcc, asynchronous host copy of find_start_padding(mfcc)) -- queued once per window's mel."""
        if mfcc is None:
            return None
        h = self._pad_handles.get(id(mfcc))
        if h is None or h[0] is not mfcc:
            h = (mfcc, _lib.HostCopy(_lib.find_start_padding(mfcc.float().reshape(1, *mfcc.shape[-2:]))))
            self._pad_handles[id(mfcc)] = h
        return h

    def hook_tokens(self, layer, ins, outs):
        cur = ins[0]
        # one decoder stream per session (T.py:806); B streams stepping together = B sessions fed by streams.py
        assert cur.shape[0] == 1, "Batch decoding is not supported"
        self.on_tokens(cur[0].tolist(), ins[0][0])    # the per-step host read whisper's own loop needs anyway

    def on_tokens(self, cur, cur_device=None):
        """One decoder call of this stream: ``cur`` = its input token ids (a list; the prompt of a new window, or the one
        token sampled at the previous step)."""
        self._commit_pending_logits()             # (REUSE_DECODER_LOGITS) the previous step's row is final by now
        tk = self.tokenizer
        sot = self._is_sot(cur)
        if sot:
            if self.language is None and len(cur) > 1:
                self.language = tk.decode(cur[-2:-1])[2:-2]      # "<|xx|>" -> "xx"
                self.opts["language"] = self.language
                self.detected_language = True
            n_sot = len(tk.sot_sequence)
            if not self.replay:                   # (only the reference-way rows need the filters)
                self.logit_filters = backend.get_logit_filters(self.model, self.opts, prompt=cur[1:-n_sot])
        self._may_flush(cur)
        if sot:
            self.has_started = len(cur) > 1 or not self.model.is_multilingual
            self.sot_index = cur.index(tk.sot) if self.no_speech_threshold is not None else self.sot_index
            assert not self.open_rows
            self.row_next = 0                     # a new window: the ring is recycled
        else:
            self.sot_index = None
        if self.has_started:
            self.segment_tokens[-1].append(cur[-1])
            self.open_rows.append(self.row_next)
            self.row_next += 1
            self.window_inputs.append(cur)
            if self.ctx_buf is not None:
                self.ctx_buf[0, self.ctx_len:self.ctx_len + len(cur)] = cur_device     # device -> device, no host list
            self.ctx_len += len(cur)
            if not sot:
                self.window_tokens_nosot.append(cur[-1])

    def text_run_is_plain(self, n):
        """Replay (streams.py): may the next n decoder calls of this stream, all of them fed one TEXT token, be booked in one
        go (`on_text_run`)?  Yes while the window stays clear of the decoding limit (T.py:878-881: no fallback token is
        recorded) -- a text token never flushes a segment (T.py:445-468 needs two timestamps in a row)."""
        if not (self.replay and self.has_started and self.reuse is True and self.window_inputs):
            return False
        after = len(self.window_tokens_nosot) + n
        return after + 2 < self.max_sample_len and after + 1 + len(self.window_inputs[0]) <= self.n_ctx

    def on_text_run(self, toks, row_in_ring):
        """`for t in toks: on_tokens([t]); hook_decoder_logits(row already in the ring)` for a run that
        `text_run_is_plain`: what those calls leave behind is bookkeeping, done here at once."""
        self._commit_pending_logits()             # the call before the run
        n = len(toks)
        self.segment_tokens[-1].extend(toks)
        self.open_rows.extend(range(self.row_next, self.row_next + n))
        self.row_next += n
        self.window_inputs.extend(toks)           # (only the first entry of a window -- its prompt -- is ever read back)
        self.ctx_len += n
        self.window_tokens_nosot.extend(toks)
        self.sot_index = None
        self.logits.n += n - 1                    # every call of the run but the last has been committed by its successor
        self.last_chunk_token = None
        self.pending_logits = (row_in_ring, False)

    def replay_prompt_logits(self, no_speech_prob):
        """Replay (streams.py): `hook_decoder_logits` for the prompt call of a window, with what it would read of the
        call's logits handed over -- the no-speech probability, computed for all streams of the loop at once by the
        backend's own decoder loop (softmax of the <|startoftranscript|> position: the expression of T.py:861-864)."""
        if self.sot_index is not None and self.no_speech_prob is None:
            self.no_speech_prob = no_speech_prob
        self.hook_decoder_logits(None, None, _ROW_IN_THE_RING)

    def hook_cross_attention(self, index, layer, ins, outs):
        assert isinstance(outs, tuple) and len(outs) == 2, "whisper seems to be outdated, please update it"
        if not self.has_started:
            return
        qk = outs[-1]
        assert qk is not None, "cross-attention QK is None: decode inside whisper.model.disable_sdpa()"
        self.ring.write(index, qk, self.open_rows[-1])

    def hook_cross_attention_fused(self, index, layer, ins, outs):
        # the last layer that owns a selected head has run: q / K of all of them are known -> ONE launch for all
        if self.has_started:
            self.ring.write_all_layers(self._q, self._k, self.open_rows[-1])
            if not self._fused_checked:
                self._check_fused_rows(self.open_rows[-1])
                for h in self._value_hooks:           # only needed for that one comparison
                    h.remove()
                self._value_hooks = []

    def _check_fused_rows(self, row):
        """Once per session: the rows wt_qk_rows computes from cross_attn.query / cross_attn.key must be the qk the
        BACKEND's own unfused attention returns for the same projections (MultiHeadAttention.qkv_attention inside
        disable_sdpa(), what the reference reads, transcribe.py:783-793) -- this is what guards the hook points, the
        head layout and the d_head ** -0.25 scaling against a backend version that does things differently."""
        self._fused_checked = True
        worst = 0.0
        with torch.no_grad(), backend.attention_weights_exposed(True):
            for index, b in enumerate(self.hooked_blocks):
                heads = self.ring._heads[index]
                if heads.numel() == 0 or self._v[index] is None:
                    continue
                ca = self.model.decoder.blocks[b].cross_attn
                out = ca.qkv_attention(self._q[index][:, -1:], self._k[index], self._v[index])
                qk = out[1] if isinstance(out, tuple) and len(out) > 1 else None
                if qk is None:          # this backend cannot show its qk: nothing to compare against
                    continue
                want = qk[0, heads.long(), -1].float()                          # (n_sel, n_ctx)
                got = self.ring.buf[self.ring._slots[index].long(), row].float()
                worst = max(worst, float((got - want).abs().max()))
        self._v = [None] * len(self.hooked_blocks)
        # (a half-precision model or a half-precision ring rounds the rows themselves: 2^-11 of |qk| ~ 10)
        q_any = next(q for q in self._q if q is not None)
        tol = 2e-3 if (q_any.dtype == torch.float32 and self.ring.buf.dtype == torch.float32) else 0.1
        if not worst <= tol:
            raise RuntimeError(f"FUSED_ATTENTION self-check failed: the QK rows computed from cross_attn.query/key differ "
                               f"from the backend's own unfused attention by {worst:.3g} (> {tol}); set "
                               f"whisper_timestamped.efficient.FUSED_ATTENTION = False for this backend")

    def hook_decoder_logits(self, layer, ins, outs):
        """REUSE_DECODER_LOGITS: forward hook on model.decoder; outs = (1, n_q, V) fp32 logits, not yet filtered."""
        if self.reuse == "auto" and self.reuse_state == "off":
            return                                    # the reference's way took over (hook_decoder_output)
        tk = self.tokenizer
        if self.sot_index is not None and self.no_speech_prob is None:
            self.no_speech_prob = outs[0, self.sot_index].float().softmax(dim=-1)[tk.no_speech].item()
        if self.language is None and self.sot_index is not None and self.model.is_multilingual:
            lo = tk.sot + 1
            probs = outs[0, self.sot_index, lo:lo + len(tk.all_language_tokens)].float().softmax(dim=-1)
            self.language_probs = dict(zip(backend.whisper().tokenizer.LANGUAGES, probs.tolist()))
        if self.has_started:
            at_limit = self.new_whisper and self._reached_decoding_limit()
            self.pending_logits = (outs, at_limit)    # a reference: the sampler filters outs[:, -1] in place

    def _commit_pending_logits(self):
        if self.pending_logits is None:
            return
        outs, at_limit = self.pending_logits
        self.pending_logits = None
        row = outs[0, -1]
        if self.reuse == "auto" and self.reuse_state == "verify":
            row = self._verified(row)
        self.logits.append(row)
        if self.reuse is True and not self.replay and len(self.logits) == 1 and self.tokenizer.no_timestamps is not None:
            # once per window: the row must carry the sampler's in-place filtering (<|notimestamps|> is always -inf)
            if not bool(torch.isinf(self.logits.buf[0, self.tokenizer.no_timestamps])):
                raise RuntimeError("REUSE_DECODER_LOGITS: this backend does not filter the decoder's logits in place; "
                                   "set whisper_timestamped.efficient.REUSE_DECODER_LOGITS = False")
        self.last_chunk_token = self.logits.argmax(-1) if at_limit else None

    def _verified(self, row):
        """First token of a window, "auto": the decoder's row as the sampler left it against the reference-way row of
        the same step (self._verify_row).  -> the row to keep."""
        want = self._verify_row
        self._verify_row = None
        if want is None:
            self.reuse_state = "off"
            return row
        want = want.reshape(-1)
        fin = torch.isfinite(want)
        tol = 1e-3 if self.embedding_t.dtype == torch.float32 else 5e-2
        same = bool(torch.equal(fin, torch.isfinite(row))) and bool(((row[fin] - want[fin]).abs() <= tol).all())
        if same:
            self.reuse_state = "trusted"
            self.stats["windows_verified"] += 1
            return row
        self.reuse_state = "off"
        self.stats["reuse_fell_back"] = True
        logger.warning("whisper_timestamped: the decoder's logits do not carry the sampler's filtering (or differ from the "
                       "re-projected ones): falling back to the reference's per-token projection + filter pass")
        return want

    def hook_decoder_output(self, layer, ins, outs):
        if self.reuse == "auto":
            if self.reuse_state == "trusted":
                return                               # the decoder's own rows are being reused
            if self.reuse_state == "verify":
                if self.has_started:                  # the reference-way row of this step, kept aside for _verified()
                    if self.embedding_t is None:
                        self.embedding_t = torch.transpose(self.model.decoder.token_embedding.weight, 0, 1).to(outs[0].dtype)
                    row = (outs[0][-1:, :] @ self.embedding_t).float()
                    context = self.ctx_buf[:, :self.ctx_len]
                    for f in self.logit_filters:
                        f.apply(row, context)
                    self._verify_row = row
                return
        tk = self.tokenizer
        if self.embedding_t is None:
            self.embedding_t = torch.transpose(self.model.decoder.token_embedding.weight, 0, 1).to(outs[0].dtype)
        x = outs[0]                               # (n_q, D)
        if self.sot_index is not None and self.no_speech_prob is None:
            row = (x[self.sot_index, :] @ self.embedding_t).float()
            self.no_speech_prob = row.softmax(dim=-1)[tk.no_speech].item()
        if self.language is None and self.sot_index is not None and self.model.is_multilingual:
            lo = tk.sot + 1
            row = (x[self.sot_index, :] @ self.embedding_t).float()
            probs = row[lo:lo + len(tk.all_language_tokens)].softmax(dim=-1)
            self.language_probs = dict(zip(backend.whisper().tokenizer.LANGUAGES, probs.tolist()))
        if self.has_started:
            logits = self.logits.next_row()                   # the step's row of the device ring: written in place
            if x.dtype == torch.float32:
                torch.matmul(x[-1:, :], self.embedding_t, out=logits)
            else:
                logits.copy_(x[-1:, :] @ self.embedding_t)     # (half model: one widening pass, as .float() was)
            context = self.ctx_buf[:, :self.ctx_len]
            for f in self.logit_filters:
                f.apply(logits, context)
            self.logits.commit()
            if self.new_whisper and self._reached_decoding_limit():
                self.last_chunk_token = self.logits.argmax(-1)
            else:
                self.last_chunk_token = None

    # ------------------------------------------------------------------ segment bookkeeping
    def _reset_open(self, add_segment, keep_last_token=True):
        if add_segment:
            if keep_last_token:
                self.segment_tokens.append([self.segment_tokens[-1][-1]])
                self.open_rows = self.open_rows[-1:]
            else:
                self.segment_tokens.append([])
                self.open_rows = []
            self.segment_tokens[-2].pop(0)        # the finished segment drops its carried-over first token
        elif len(self.segment_tokens[-1]) > 0:
            self.segment_tokens[-1] = []
            self.open_rows = []

    def _must_flush(self, cur):
        tk = self.tokenizer
        open_seg = self.segment_tokens[-1]
        if not self._is_sot(cur):
            both = self._is_ts(cur[0]) and bool(open_seg) and self._is_ts(open_seg[-1])
            if both:
                self.saw_consecutive_timestamps = True
            return both
        flush = len(open_seg) > 1 and not self.saw_consecutive_timestamps
        if not flush and self.new_whisper:        # the open segment ends with a timestamp: it is a real segment
            if self.last_chunk_token is None:
                flush = len(open_seg) > 2 and self._is_ts(open_seg[-1])
            else:
                flush = self._is_ts(self.last_chunk_token)
        if not flush and self.trust:
            self._reset_open(False)               # drop the unfinished tail of the window
        self.saw_consecutive_timestamps = False
        return flush

    def _window_start_index(self, cur):
        if self._is_sot(cur) and self.has_started:
            if self.trust:
                res = self.first_segment_of_window
                self.first_segment_of_window = len(self.segment_tokens) - 1
            else:
                res = len(self.segment_tokens) - 1
            return res
        return None

    def _start_of_padding(self):
        ref, val = self._pad_cache
        if ref is not self.mfcc:
            val = None
            if self.mfcc is not None:
                m = self.mfcc
                sp = int(_lib.find_start_padding(m.float().reshape(1, *m.shape[-2:]))[0])
                val = None if sp < 0 else sp
            self._pad_cache = (self.mfcc, val)
        return val

    def _align_open_segment(self, cur=None):
        tk = self.tokenizer
        tokens = self.segment_tokens[-1][1:]
        unfinished = self._reached_decoding_limit()
        last_not_ts = bool(tokens) and not self._is_ts(tokens[-1])
        reliable = True
        if unfinished:                            # the decoder hit its token budget: recover the last sampled token
            if cur is not None and cur[0] == tk.sot_prev:
                pos = [i for i, t in enumerate(cur) if t == tk.sot]
                assert len(pos) == 1 and pos[0] > 0
                self.last_token_fallback = cur[pos[0] - 1]      # it is the end of the next window's prompt
            else:
                self.last_token_fallback = self.logits.argmax(-1) if self.last_chunk_token is None else self.last_chunk_token
                reliable = (self.temperature == 0)
            tokens.append(self.last_token_fallback)
            self.segment_tokens[-1].append(self.last_token_fallback)
            rows = list(self.open_rows)
            last_row = -1
        elif last_not_ts:                         # <|endoftext|> came without a closing timestamp
            tokens.append(tk.eot)
            self.segment_tokens[-1].append(tk.eot)
            rows = list(self.open_rows)
            last_row = -1
        else:
            rows = list(self.open_rows[:-1])
            last_row = -2

        end_token = tokens[-1]
        if self._is_ts(end_token):
            start_token = tokens[0]
            assert self._is_ts(start_token)
            if end_token <= start_token:          # obviously wrong end: constrained re-estimate after the start
                tokens[-1] = self.logits.argmax(last_row, lo=start_token + 1)

        if len(tokens) <= 1:
            ws = []
        elif self.defer:
            unit = prepare_unit(tokens, None, tk, use_space=backend.should_use_space(self.language),
                                refine_whisper_precision_nframes=self.refine_nframes,
                                remove_punctuation_from_words=self.remove_punctuation_from_words,
                                unfinished_decoding=unfinished, detect_disfluencies=False,
                                start_of_padding=None, qk_selected=self.ring.rows(rows))
            if unit is None:
                ws = []
            else:
                # texts and tokens of the words are known now; their times arrive with the window's batch
                ws = [dict(text=text, start=None, end=None, tokens=pieces, tokens_indices=ids)
                      for text, pieces, ids in planned_words(unit, with_text=True)]
                if ws:
                    if self.plot:
                        unit.mel = self.mfcc
                    self.queued.append((unit, ws, self._padding_handle(self.mfcc)))
        else:
            unit = prepare_unit(tokens, None, tk, use_space=backend.should_use_space(self.language),
                                refine_whisper_precision_nframes=self.refine_nframes,
                                remove_punctuation_from_words=self.remove_punctuation_from_words,
                                unfinished_decoding=unfinished, detect_disfluencies=self.detect_disfluencies,
                                start_of_padding=self._start_of_padding(), qk_selected=self.ring.rows(rows))
            if unit is None:
                ws = []
            else:
                batch = AlignmentBatch(workspace=self.workspace, plot=self.plot)
                if self.plot:
                    unit.mel = self.mfcc
                batch.add(unit)
                ws = batch.run()[0]
        added = len(ws) > 0
        if added:
            self.words_per_segment.append(ws)
        self._reset_open(added, not self._is_sot(cur))
        return added, unfinished, reliable

    # ------------------------------------------------------------------ deferred alignment
    def _launch_queued(self):
        """The window is closed: ONE launch set for all of its segments (their QK rows are still in the ring: the
        kernels are queued on the stream before the next window's tokens overwrite them), then read the PREVIOUS
        window's record -- that copy was queued a whole window ago, the wait does not stall anything."""
        if self.sink is not None:              # B streams: the driver launches ONE set for every stream's units
            self.sink.take(self.queued)
            self.queued = []
            return
        previous, self.in_flight = self.in_flight, []
        if self.queued:
            batch = AlignmentBatch(workspace=self.workspace, plot=self.plot)
            for unit, _, handle in self.queued:
                if handle is not None:
                    sp = int(handle[1].wait()[0])
                    set_padding(unit, None if sp < 0 else sp)
                batch.add(unit)
            batch.launch().fetch()
            self.stats["alignment_launch_sets"] += 1
            self.in_flight.append((batch, [ws for _, ws, _ in self.queued]))
            self.queued = []
        self._collect(previous)

    @staticmethod
    def _collect(batches):
        for batch, placeholders in batches:
            for ws, real in zip(placeholders, batch.collect()):
                assert len(ws) == len(real), f"planned {len(ws)} words, aligned {len(real)}"
                for w, r in zip(ws, real):
                    assert w["tokens_indices"] == r["tokens_indices"]
                    w["start"], w["end"] = r["start"], r["end"]

    def _resolve_all(self):
        if self.sink is not None:
            self._launch_queued()
            self.sink.resolve()
            return
        self._launch_queued()
        pending, self.in_flight = self.in_flight, []
        self._collect(pending)

    # ------------------------------------------------------------------ flush logic
    def _may_flush(self, cur=None):
        unfinished, reliable = False, True
        if self._must_flush(cur) and self.trust:
            _, unfinished, reliable = self._align_open_segment(cur)
        i_start = self._window_start_index(cur)
        if i_start is None:
            return
        if not self.trust:
            unfinished, reliable = self._align_whole_window(unfinished, reliable)
        if self.defer:
            self._launch_queued()
        self.mfcc = self.new_mfcc
        self._close_window(i_start, unfinished, reliable)

    def _align_whole_window(self, unfinished, reliable):
        """trust_whisper_timestamps=False: one alignment for the whole 30 s window, then re-split at the
        consecutive timestamp pairs (transcribe.py:586-706)."""
        tk = self.tokenizer
        ts0 = tk.timestamp_begin
        toks = list(self.segment_tokens[-1])
        n = len(toks)
        idx_task = toks.index(tk.sot_sequence[-1])
        special = [t >= tk.eot for t in toks]
        for i in range(idx_task):
            special[i] = True                     # prompt
        for i in range(idx_task, min(idx_task + 2, n)):
            special[i] = False                    # task token slot + begin timestamp
        is_ts = [t >= ts0 for t in toks]
        consecutive = [i for i in range(n - 1) if is_ts[i + 1] and is_ts[i]]
        if (self.new_whisper or self._reached_decoding_limit()) and (
                (is_ts[-1] and not is_ts[-2]) if self.last_chunk_token is None
                else (self.last_chunk_token >= ts0 and not is_ts[-2])):
            consecutive.append(n - 1)
        last_is_ts = True
        if consecutive:
            for i in range(consecutive[-1] + 1, n):
                special[i] = True
            special[consecutive[-1]] = False
        elif is_ts[-1]:
            special[-1] = False
        else:
            last_is_ts = False
        if self.use_timestamps_for_alignment and consecutive:
            for i in range(idx_task + 2, consecutive[-1]):
                special[i] = False
        next_kept = [not s for s in special[1:]] + [True]
        assert len(self.open_rows) == n, f"{len(self.open_rows)} attention weights != {n}"
        self.open_rows = [r for keep, r in zip(next_kept, self.open_rows) if keep]     # rows that predicted a kept token
        kept = [t for t, s in zip(toks, special) if not s]
        assert len(self.open_rows) == len(kept), f"{len(self.open_rows)} attention weights != {len(kept)} "
        orig_start, orig_end = kept[1], kept[-1]
        kept[1] = ts0
        if last_is_ts:
            kept[-1] = ts0 + N_AUDIO_CTX
        self.segment_tokens[-1] = kept

        added, unfinished, reliable = self._align_open_segment()
        if added and self.defer and not self.use_timestamps_for_alignment:
            self._resolve_all()                   # (this branch reads word times while re-splitting)
        if added:
            if len(consecutive) > 1:
                concat = self.words_per_segment[-1]
                new_words, new_tokens = [], []
                start = idx_task + 1
                i_word = 0
                for i, end in enumerate(consecutive):
                    new_tokens.append(toks[start:end + 1])
                    total = end - start - 1
                    start = end + 1
                    length = 0
                    new_words.append([])
                    while length < total:
                        if not self.use_timestamps_for_alignment and i_word == len(concat):
                            assert total == 1 and i == len(consecutive) - 1, "Unexpected situation!"
                            break
                        assert i_word < len(concat), f"i_word={i_word} < len(segments_timestamped_concat)={len(concat)}"
                        word = concat[i_word]
                        new_words[-1].append(word)
                        length += len(word["tokens_indices"])
                        i_word += 1
                    if self.use_timestamps_for_alignment:
                        assert length == total, f"length={length} != total_length={total}"
                    elif length > total:          # a lone punctuation glued onto the previous segment: split it back
                        delta = length - total
                        word = new_words[-1][-1]
                        ids, pieces = word["tokens_indices"], word["tokens"]
                        word["tokens_indices"], word["tokens"] = ids[:-delta], pieces[:-delta]
                        word["word"] = "".join(pieces[:-delta])
                        i_word -= 1
                        t = concat[i_word]["end"]
                        concat[i_word] = dict(text="".join(pieces[-delta:]), start=t, end=t, tokens=pieces[-delta:],
                                              tokens_indices=ids[-delta:])
                assert i_word == len(concat)
                self.segment_tokens = self.segment_tokens[:-2] + new_tokens + [self.segment_tokens[-1]]
                self.words_per_segment = self.words_per_segment[:-1] + new_words
            else:
                seg = self.segment_tokens[-2]
                seg[0] = orig_start
                if last_is_ts:
                    seg[-1] = orig_end
            if unfinished:
                self.words_per_segment[-1][-1]["avg_logprob_reliable"] = reliable
        self._reset_open(False)
        return unfinished, reliable

    def _close_window(self, i_start, unfinished, reliable):
        """All segments of the previous 30 s window are known: confidences, retroactive no-speech skip
        (transcribe.py:708-781)."""
        tk = self.tokenizer
        n_segments = len(self.segment_tokens) - 1
        skip = False
        if self.compute_word_confidence or self.no_speech_threshold is not None:
            skip = (self.no_speech_prob > self.no_speech_threshold) if self.no_speech_threshold is not None else False
            avg_logprob = None
            if self.compute_word_confidence or (skip and self.logprob_threshold is not None):
                n = len(self.logits)
                if n == len(self.window_tokens_nosot):
                    self.window_tokens_nosot = self.window_tokens_nosot[1:]
                if unfinished:
                    assert self.last_token_fallback is not None
                    last = [self.last_token_fallback]
                    self.words_per_segment[-1][-1]["avg_logprob_reliable"] = reliable
                    n += 1
                elif self._reached_decoding_limit():    # segments were found, then the language model got stuck
                    last = [self.logits.argmax(-1)]
                    self.words_per_segment[-1][-1]["avg_logprob_reliable"] = (self.temperature == 0)
                else:
                    last = [tk.eot]
                chosen = self.window_tokens_nosot + last
                assert len(self.logits) == len(chosen), f"{len(self.logits)} != {len(chosen)}"
                logprobs = self.logits.gather(chosen).cpu()            # ONE kernel + one KB-sized copy per window
                assert bool(torch.isfinite(logprobs).all()), \
                    f"Got infinite logprob among ({len(logprobs)}) {list(zip(chosen, logprobs.tolist()))}"
                total = np.float32(0)
                for v in logprobs.numpy():                              # sequential fp32 sum, like sum(tensors)
                    total = np.float32(total + v)
                avg_logprob = float(np.float32(total / np.float32(n)))
                if self.logprob_threshold is not None and avg_logprob > self.logprob_threshold:
                    skip = False
            if skip:
                self.first_segment_of_window -= n_segments - i_start
                self.segment_tokens = self.segment_tokens[:i_start] + [self.segment_tokens[-1]]
                self.words_per_segment = self.words_per_segment[:i_start]
            elif self.compute_word_confidence:
                i_end = -1
                for i in range(i_start, n_segments):
                    tokens = self.segment_tokens[i]
                    i_begin = i_end + 1
                    i_end = i_begin + len(tokens)
                    assert chosen[i_begin:i_end] == tokens, \
                        f"Inconsistent token list {tk.decode_with_timestamps(chosen[i_begin:i_end])} != {tk.decode_with_timestamps(tokens)}"
                    i_begin += 1                                        # skip the start timestamp
                    if not unfinished or i != n_segments - 1:
                        i_end -= 1                                      # skip the end timestamp
                    self.segment_logprobs.append(logprobs[i_begin:i_end])
                    self.segment_avglogprobs.append(avg_logprob)
            else:
                for _ in range(i_start, n_segments):
                    self.segment_logprobs.append(None)
                    self.segment_avglogprobs.append(None)
        else:
            for _ in range(i_start, n_segments):
                self.segment_logprobs.append(None)
                self.segment_avglogprobs.append(None)
        self.window_inputs = []
        self.ctx_len = 0
        self.window_tokens_nosot = []
        self.logits.reset()
        self.no_speech_prob = None
        if self.reuse_state == "trusted":
            self.reuse_state = "verify"          # every window proves itself on its first token

    # ------------------------------------------------------------------ driver
    def run(self, audio):
        model = self.model
        hooks = [model.encoder.conv1.register_forward_hook(self.hook_mel),
                 model.decoder.token_embedding.register_forward_hook(self.hook_tokens)]
        try:
            used = self.ring.used_layers()               # layers that own a selected head: the only ones observed
            for j, b in enumerate(self.hooked_blocks):
                ca = model.decoder.blocks[b].cross_attn
                if FUSED_ATTENTION:
                    if j not in used:
                        continue
                    hooks.append(ca.query.register_forward_hook(lambda m, i, o, index=j: self._q.__setitem__(index, o)))
                    hooks.append(ca.key.register_forward_hook(lambda m, i, o, index=j: self._k.__setitem__(index, o)))
                    self._value_hooks.append(ca.value.register_forward_hook(
                        lambda m, i, o, index=j: self._v.__setitem__(index, o)))
                    if j == used[-1]:
                        hooks.append(ca.register_forward_hook(
                            lambda layer, ins, outs, index=j: self.hook_cross_attention_fused(index, layer, ins, outs)))
                else:
                    hooks.append(ca.register_forward_hook(
                        lambda layer, ins, outs, index=j: self.hook_cross_attention(index, layer, ins, outs)))
            if self.compute_word_confidence or self.no_speech_threshold is not None:
                if self.reuse in (True, "auto"):
                    hooks.append(model.decoder.register_forward_hook(self.hook_decoder_logits))
                if self.reuse in (False, "auto"):
                    hooks.append(model.decoder.ln.register_forward_hook(self.hook_decoder_output))
            with torch.no_grad(), backend.attention_weights_exposed(not FUSED_ATTENTION), \
                    backend.gpu_log_mel(model.device, GPU_FRONT_END):
                transcription = model.transcribe(audio, **self.opts)
        finally:
            for h in hooks + self._value_hooks:
                h.remove()
            self._value_hooks = []
        self.end_of_stream()
        if self.defer:
            self._resolve_all()                   # the last window's record
        return self.compiled(transcription)

    def end_of_stream(self):
        """The backend has returned: the last decoder call's row, the last open segment (T.py:913)."""
        self._commit_pending_logits()
        self._may_flush()

    def compiled(self, transcription):
        self.segment_tokens.pop(-1)
        LAST_SESSION.clear()
        LAST_SESSION.update(self.stats, reuse_state=self.reuse_state)
        return self._compile(transcription)

    def _compile(self, transcription):
        """Reconcile with whisper's own segment list, add offsets and confidences (transcribe.py:916-1001)."""
        tk = self.tokenizer
        special0 = min(tk.sot, tk.eot)

        def strip_special(tokens):
            a, b = 0, len(tokens)
            while a < b and tokens[a] >= special0:
                a += 1
            while b > a and tokens[b - 1] >= special0:
                b -= 1
            return tokens[a:b]

        n_tok, n_words = len(self.segment_tokens), len(self.words_per_segment)
        assert n_tok == n_words, f"Inconsistent number of segments: tokens ({n_tok}) != timestamped_word_segments ({n_words})"
        assert len(self.segment_avglogprobs) == n_tok, \
            f"Inconsistent number of segments: avg logprobs ({len(self.segment_avglogprobs)}) != tokens ({n_tok})"
        assert len(self.segment_logprobs) == n_tok, \
            f"Inconsistent number of segments: logprobs ({len(self.segment_logprobs)}) != tokens ({n_tok})"
        segments = transcription["segments"]
        if any(not s["text"] for s in segments):
            segments = [s for s in segments if s["text"]]
        l1, l2 = len(segments), n_words
        if l1 != l2 and l1 != 0:
            logger.warning(f"Inconsistent number of segments: whisper_segments ({l1}) != timestamped_word_segments ({l2})")
        assert l1 == l2 or l1 == 0, f"Inconsistent number of segments: whisper_segments ({l1}) != timestamped_word_segments ({l2})"

        words = []
        for i, (segment, seg_words, token, avglogprob, logprobs) in enumerate(
                zip(segments, self.words_per_segment, self.segment_tokens, self.segment_avglogprobs, self.segment_logprobs)):
            mine, theirs = strip_special(token), strip_special(segment["tokens"])
            if mine != theirs:
                if len(mine) == len(theirs) + 1:
                    logger.warning(f"An additional token was added on segment {i}")
                elif self.new_whisper and len(theirs) == 0:
                    logger.warning(f"Whisper has empty segment {i}")
                    assert segment["end"] == segment["start"], f"Fatal Error: Got empty segment {i} with non-zero duration"
                    segment["tokens"] = mine
                    segment["text"] = tk.decode(mine)
                else:
                    assert len(mine) < len(theirs) and mine == theirs[:len(mine)], \
                        f"Fatal Error: Got inconsistent text for segment {i}:\n({len(mine)})\n{mine}\n!=\n({len(theirs)})\n{theirs[:len(mine)]}"
                    segment["tokens"] = token if self.new_whisper else mine
                    segment["text"] = tk.decode(segment["tokens"])
                    logger.warning(f"Text had to be shortned on segment {i}:\n{tk.decode(mine)}\n!=\n{tk.decode(theirs)}")
                seg_words[-1]["avg_logprob_reliable"] = False

            offset = segment["seek"] * HOP_LENGTH / SAMPLE_RATE
            for w in seg_words:
                w["start"] += offset
                w["end"] += offset
                w["idx_segment"] = i

            if self.compute_word_confidence:
                if seg_words[-1].get("avg_logprob_reliable", True) and abs(segment["avg_logprob"] - avglogprob) >= 1e-2:
                    logger.warning(f"Recomputed different logprob for segment {i}: {avglogprob} != {segment['avg_logprob']}")
                conf, consumed = segment_confidences(seg_words, logprobs, self.include_punctuation_in_confidence)
                segment["confidence"] = conf
                if consumed not in (len(logprobs), len(logprobs) - 1):
                    logger.warning(f"Got inconsistent length for segment {i} ({len(logprobs)} != {consumed}). Some words have been ignored.")
            words.extend(seg_words)

        if self.language_probs:
            transcription["language_probs"] = self.language_probs
        return transcription, words


def transcribe_efficient(model, audio, *, remove_punctuation_from_words, compute_word_confidence,
                         include_punctuation_in_confidence, refine_whisper_precision_nframes, alignment_heads,
                         plot_word_alignment, word_alignment_most_top_layers, detect_disfluencies,
                         trust_whisper_timestamps, use_timestamps_for_alignment=True, **whisper_options):
    verbose = whisper_options["verbose"]
    whisper_options["verbose"] = None if verbose is True else verbose     # words are printed by the caller
    if verbose and whisper_options["language"] is None and getattr(model, "is_multilingual", False):
        print("Detecting language using up to the first 30 seconds. Use `--language` to specify the language")
    session = EfficientSession(model, whisper_options, remove_punctuation_from_words=remove_punctuation_from_words,
                               compute_word_confidence=compute_word_confidence,
                               include_punctuation_in_confidence=include_punctuation_in_confidence,
                               refine_whisper_precision_nframes=refine_whisper_precision_nframes,
                               alignment_heads=alignment_heads,
                               word_alignment_most_top_layers=word_alignment_most_top_layers,
                               detect_disfluencies=detect_disfluencies, trust_whisper_timestamps=trust_whisper_timestamps,
                               use_timestamps_for_alignment=use_timestamps_for_alignment, plot=plot_word_alignment)
    out = session.run(audio)
    if verbose and session.detected_language:
        print(f"Detected language: {backend.whisper().tokenizer.LANGUAGES[session.language].title()}")
        sys.stdout.flush()
    return out
