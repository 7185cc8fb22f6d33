"""Batched form of the naive strategy's second pass: many independent 30 s windows
through log-mel, encoder, teacher-forced decoder, QK-row capture, word alignment
and confidence gather AT ONCE.

The reference re-runs the model window by window
(/root/reference/whisper_timestamped/transcribe.py:1131-1323): for each window
one ``log_mel_spectrogram`` (:1211-1215), one ``model(mfcc, tokens)`` with the
attention hooks (:1236-1245), one ``perform_word_alignment`` (:1251-1262) and a
Python loop of ``logprobs[:, step, tok]`` reads (:1285-1300).  With
``trust_whisper_timestamps=False`` the windows are whisper's 30 s seek groups and
do not depend on each other (:1197-1202) -- the ``previous_end`` chain only
exists in the other branch (:1147-1152) -- so B of them can share every launch:

    PCM (B, 480000) --wt_logmel_pad_batch--> mel (B, n_mels, 3000), pad[B] (find_start_padding, a 3 us pass behind it)
    model.encoder / model.decoder on the whole batch (torch: hipBLASLt GEMMs, fused attention)
    cross_attn.query / cross_attn.key outputs of every hooked layer --wt_qk_rows_batch--> rows (B, A, T_max, 1500)
    ONE wt_align_batch over all windows' units --> jumps
    ONE wt_logprob_gather_rows over all windows' text positions --> log-probabilities
    ONE device->host copy of (jumps | log-probs), then the reference's host-side word assembly per window.

``align_windows`` is that pipeline; ``naive.transcribe_naive`` hands it the windows of one recording,
``bench.py`` 32 synthetic chunks (BASELINE.json configs[1] at the transcribe() level).  Sub-batches are pipelined:
the GPU work of sub-batch k+1 is queued before the host assembles the words of sub-batch k.

Streams (``SCHEDULE``; whisper_timestamped/pipeline.py): the model's GEMMs, the QK-row pass, the cost stage and the
log-probability gather stay on the caller's current stream, in order; the log-mel front end and the DTW of sub-batch k go
to high-priority HIP stream k % 2 of the aligner's own, dependencies as events (pipeline.StageSet) -- so the latency-bound
DTW of sub-batch k and the VALU-bound STFT of sub-batch k + 1 run beside the other kernels instead of in front of them.
``SCHEDULE = "serial"`` (the default: see below) keeps everything on the caller's current stream; results are
bit-identical either way (tests/test_gpu_transcribe.py).
"""
from __future__ import annotations

import logging
from dataclasses import dataclass, field

import numpy as np
import torch

from . import _lib, audio as wt_audio, backend, pipeline
from .alignment import AlignmentBatch, default_workspace, head_pairs, planned_words, prepare_unit, set_padding
from .capture import layer_head_slots
from .confidence import strip_trailing_punctuation
from .words import AUDIO_SAMPLES_PER_TOKEN, HOP_LENGTH, N_FRAMES

logger = logging.getLogger("whisper_timestamped")

N_SAMPLES = N_FRAMES * HOP_LENGTH     # 480000: one 30 s window
MAX_WINDOWS_PER_LAUNCH = 32           # BASELINE.json configs[1]; bounds the (B, T_max, V) logits block (base: 0.6 GB)
# "serial" (default) = every stage on the caller's current stream; "hilo" = the front end and the DTW on high-priority
# streams of the aligner's own (pipeline.StageSet).  Measured on the fp32 whisper-base leg, 32 windows per launch set
# (profiles/r6d_batched_hang_variants.txt): hilo 23.59-23.68 k audio-s/s, serial 23.77 k -- the model's GEMMs are 99 % of the
# GPU time there, the alignment stages have nothing to hide behind -- so the default stays serial; hilo is kept, tested
# bit-identical, for models whose forward pass is cheap next to the alignment (tiny models, half precision).
SCHEDULE = "serial"


@dataclass
class WindowJob:
    """One independent window = one iteration of the reference's loop body (:1204-1323)."""
    pcm: torch.Tensor               # 1-D crop of the recording (<= 30 s; already minimum-padded), any device
    tokens: list                    # the window's tokens as the loop hands them over (timestamps at both ends allowed)
    n_crop_samples: int             # end_sample - start_sample of the crop (before the minimum padding, :1240)
    tag: object = None


@dataclass
class WindowResult:
    words: list                     # perform_word_alignment's output for the window (times relative to the crop)
    word_logprobs: list             # per word: CPU fp32 tensor of its kept tokens' log-probs (None without confidence)
    tokens: list                    # tokens[i_start:] + [end_token] (what the alignment saw, :1241)
    first_token_check: int | None   # leading timestamp that was stripped (:1219-1221)
    last_token_check: int | None    # trailing timestamp that was stripped (:1225-1228)
    tag: object = None


@dataclass
class _Stage:                       # one sub-batch in flight
    jobs: list
    batch: AlignmentBatch = None
    units: list = field(default_factory=list)          # per job: AlignmentUnit | None
    plans: list = field(default_factory=list)          # per job: [(offset into extra, n)] per planned word
    tokens: list = field(default_factory=list)
    checks: list = field(default_factory=list)
    keep: list = field(default_factory=list)           # tensors that must outlive the asynchronous launches
    marks: list = field(default_factory=list)          # (stage name, event) when the aligner keeps a timeline


class BatchedAligner:
    """Everything of ``align_windows`` that is per model: hooks, head tables, workspaces."""

    def __init__(self, model, tokenizer, *, language, use_space=None, alignment_heads="model",
                 word_alignment_most_top_layers=None, refine_whisper_precision_nframes=25,
                 remove_punctuation_from_words=False, compute_word_confidence=True,
                 include_punctuation_in_confidence=False, detect_disfluencies=False, ring_dtype=torch.float32,
                 mel_dtype=None, fused_attention=None):
        from . import efficient
        self.model, self.tk = model, tokenizer
        self.dev = model.device
        _lib.require_gpu(self.dev, "the batched window aligner")
        self.language = language
        self.use_space = backend.should_use_space(language) if use_space is None else use_space
        if isinstance(alignment_heads, str):
            alignment_heads = getattr(model, "alignment_heads", None)
        n_blocks = len(model.decoder.blocks)
        top = n_blocks if word_alignment_most_top_layers is None else min(word_alignment_most_top_layers, n_blocks)
        self.hooked = list(range(n_blocks - top, n_blocks))
        self.n_heads = model.dims.n_text_head
        self.n_ctx = model.dims.n_audio_ctx
        self.pairs = head_pairs(alignment_heads)
        per_layer, self.n_slots = layer_head_slots(self.pairs, len(self.hooked), self.n_heads)
        # only the layers that own a selected head are observed (whisper-small: 4 of 12, large-v3: 10 of 32 -- every
        # observed layer keeps its (B, 1500, D) key projection alive until the rows are computed)
        self.used = [l for l, (hs, _) in enumerate(per_layer) if hs]
        sel = [(i, h, s) for i, l in enumerate(self.used) for h, s in zip(*per_layer[l])]
        self.sel_layer, self.sel_head, self.sel_slot = (
            torch.tensor([x[i] for x in sel], dtype=torch.int32, device=self.dev) for i in range(3))
        self.refine, self.remove_punct = refine_whisper_precision_nframes, remove_punctuation_from_words
        self.want_conf, self.incl_punct = compute_word_confidence, include_punctuation_in_confidence
        self.disfl = detect_disfluencies
        self.ring_dtype, self.mel_dtype = ring_dtype, mel_dtype
        self.fused = efficient.FUSED_ATTENTION if fused_attention is None else fused_attention
        self.n_mels = model.dims.n_mels if hasattr(model.dims, "n_mels") else 80
        self.workspace = default_workspace(self.dev)
        self.timeline = None            # set to [] to collect per-sub-batch GPU stage times (ms) -- bench.py does
        self.schedule = pipeline.choose_schedule(SCHEDULE, MAX_WINDOWS_PER_LAUNCH)     # ("auto" = the pipeline's rule: hilo)
        self._stage_sets, self._launches = None, 0
        # (Rounds 2-3 carried an opt-in replay of the forward pass as one captured HIP graph per shape.  Since round 3 the
        #  eager half-precision pass is GPU-bound -- 96 % busy in its own process -- so a replay could gain 4 % at most, and
        #  it measured 1.4-2x SLOWER, 33-48 k against 66 k audio-s/s: removed in round 4, docs/history/DESIGN_rounds_1-4.md section 6a.)
        sot = tokenizer.sot_sequence
        if language and len(sot) == 3:                                   # :1230-1232
            sot = (sot[0], tokenizer.to_language_token(language), sot[2])
        self.sot_sequence = tuple(sot)

    # ------------------------------------------------------------------ host: token preparation (:1217-1234)
    def _prepare_tokens(self, tokens):
        ts0 = self.tk.timestamp_begin
        tokens = list(tokens)
        first = tokens[0] if tokens[0] >= ts0 else None
        while tokens[0] >= ts0:
            tokens = tokens[1:]
            assert len(tokens), "Got transcription with only timestamps!"
        last = None
        while tokens[-1] >= ts0:
            last = tokens[-1]
            tokens = tokens[:-1]
        return [*self.sot_sequence, ts0] + tokens, first, last

    def _to_device(self, array, keep):
        """Small int32 host array -> device through pooled pinned memory, asynchronously (no stream synchronisation)."""
        up = _lib.PinnedUpload(np.ascontiguousarray(array, dtype=np.int32), self.dev)
        keep.append(up)
        return up.dev

    def _mark(self, st, name, stream=None):
        """Timeline (bench.py): an event pair around every stage's launches -- the pair times the stage's kernels only,
        not the time the GPU may have waited for the host to queue them."""
        if self.timeline is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(stream if stream is not None else torch.cuda.current_stream(self.dev))
            st.marks.append((name, ev))

    def _next_stage_set(self):
        """StageSet of the next sub-batch, or None under the serial schedule: the caller's CURRENT stream as the set's
        low-priority lane (everything torch launches -- the model's GEMMs, the QK-row pass, the cost stage, the gather --
        stays on one stream, in order, sub-batch after sub-batch) and one of two high-priority streams of the aligner's own,
        alternating, for the log-mel front end and the DTW.  Two sub-batches' GEMMs never run side by side: with a
        low-priority stream per sub-batch the fp32 model's forward passes of two sub-batches overlapped and the device
        hung in the backend's GEMM kernels, nondeterministically (profiles/r6d_batched_hang_variants.txt)."""
        if self.schedule != "hilo":
            return None
        with _lib.device_ctx(self.dev):
            if self._stage_sets is None:
                self._stage_sets = [pipeline.priority_stream(self.dev, "high") for _ in range(2)]
            self._launches += 1
            return pipeline.StageSet(self.dev, "hilo", streams=(self._stage_sets[self._launches % 2],
                                                                 torch.cuda.current_stream(self.dev)))

    def close(self):
        """Free the library's scratch arenas of the aligner's own streams (after the last collect())."""
        if self._stage_sets is not None:
            with _lib.device_ctx(self.dev):
                for hi in self._stage_sets:
                    hi.synchronize()
                    _lib.release_stream(hi)
            self._stage_sets = None

    # ------------------------------------------------------------------ the model's forward pass
    def _forward(self, x, tok_dev):
        """-> (logits (B, T, V), q_out, k_out, captured): eager, with the capture hooks installed for the call."""
        n = len(self.used)
        q_out, k_out, captured = [None] * n, [None] * n, [None] * n
        hooks = []
        try:
            for j, l in enumerate(self.used):
                ca = self.model.decoder.blocks[self.hooked[l]].cross_attn
                if self.fused:
                    hooks.append(ca.query.register_forward_hook(lambda m, i, o, j=j: q_out.__setitem__(j, o)))
                    hooks.append(ca.key.register_forward_hook(lambda m, i, o, j=j: k_out.__setitem__(j, o)))
                else:
                    hooks.append(ca.register_forward_hook(lambda m, i, o, j=j: captured.__setitem__(j, o[1])))
            with backend.attention_weights_exposed(not self.fused):
                logits = self.model(x, tok_dev)                                  # (B, T_max, V) fp32
        finally:
            for h in hooks:
                h.remove()
        return logits, q_out, k_out, captured

    # ------------------------------------------------------------------ device: one sub-batch, nothing waits
    def launch(self, jobs) -> _Stage:
        return self._launch(jobs, self._next_stage_set())

    def _launch(self, jobs, ss) -> _Stage:
        tk, dev = self.tk, self.dev
        st = _Stage(jobs=list(jobs))
        B = len(st.jobs)
        i_start = len(self.sot_sequence)
        fed = []
        for job in st.jobs:
            toks, first, last = self._prepare_tokens(job.tokens)
            fed.append(toks)
            st.checks.append((first, last))
        T_max = max(len(t) for t in fed)
        assert T_max <= self.model.dims.n_text_ctx, f"{T_max} tokens do not fit the decoder's context"
        tok_mat = np.full((B, T_max), tk.eot, dtype=np.int32)               # right padding: causal attention ignores it
        n_valid = np.zeros(B, dtype=np.int32)
        for b, (job, toks) in enumerate(zip(st.jobs, fed)):
            tok_mat[b, :len(toks)] = toks
            n = int(job.pcm.shape[-1])
            assert n <= N_SAMPLES, f"window {b}: {n} samples > 30 s (the batched path takes whisper's 30 s seek groups)"
            n_valid[b] = n
        with _lib.device_ctx(dev), torch.no_grad():
            def stage_pcm():
                pcm_ = torch.zeros((B, N_SAMPLES), dtype=torch.float32, device=dev)
                for b_, job_ in enumerate(st.jobs):
                    pcm_[b_, :n_valid[b_]].copy_(job_.pcm.reshape(-1), non_blocking=True)
                return pcm_
            # log-mel of every crop, zero padded to 3000 frames (:1211-1215), and where the padding starts (:1795-1805)
            if ss is None:
                pcm = stage_pcm()
                small = self._to_device(np.concatenate([tok_mat.reshape(-1), n_valid]), st.keep)
                tok_dev, nv_dev = small[:B * T_max].view(B, T_max), small[B * T_max:]
                self._mark(st, "logmel<")
                mel, pad = wt_audio.log_mel_batch(pcm, nv_dev, n_mels=self.n_mels, n_frames=N_FRAMES, with_padding=True)
                pad_copy = _lib.HostCopy(pad)
                self._mark(st, "logmel>")
                del pcm
            else:
                # The whole front end -- staging the crops, the STFT, the padding index and its copy to the host -- on the
                # set's high-priority stream: queued NOW, beside whatever the previous sub-batch still has on the caller's
                # stream; the model waits for it by event.  Buffers allocated under that stream and read on the caller's
                # (mel) are handed over with record_stream.
                cur = torch.cuda.current_stream(dev)
                ss.hi.wait_stream(cur)                                  # the jobs' PCM was produced on the caller's stream
                with torch.cuda.stream(ss.hi):
                    pcm = stage_pcm()
                    nv_dev = self._to_device(n_valid, st.keep)
                    self._mark(st, "logmel<", ss.hi)
                    mel, pad = wt_audio.log_mel_batch(pcm, nv_dev, n_mels=self.n_mels, n_frames=N_FRAMES, with_padding=True,
                                                      launch=lambda fn: ss.run("logmel", fn))
                    pad_copy = _lib.HostCopy(pad)
                    self._mark(st, "logmel>", ss.hi)
                    del pcm
                mel.record_stream(cur)
                ss.wait_for(cur, ("logmel",))
                tok_dev = self._to_device(tok_mat.reshape(-1), st.keep).view(B, T_max)
            self._mark(st, "model<")
            # encoder + teacher-forced decoder on the whole batch (:1236-1238); no logit filters on this path (:1245)
            x = mel if self.mel_dtype is None else mel.to(self.mel_dtype)
            logits, q_out, k_out, captured = self._forward(x, tok_dev)
            self._mark(st, "model>")
            self._mark(st, "qk_rows<")
            # the alignment heads' QK rows of every window: (B, A, T_max, 1500)
            ring = torch.empty((B, self.n_slots, T_max, self.n_ctx), dtype=self.ring_dtype, device=dev)
            lens = np.array([len(t) for t in fed], dtype=np.int32)
            if self.fused:
                rows = self._to_device(np.concatenate([np.full(B, i_start - 1, dtype=np.int32), lens]), st.keep)
                _lib.qk_rows_batch([q.contiguous() for q in q_out], [k.contiguous() for k in k_out], self.sel_layer,
                                   self.sel_head, self.sel_slot, ring, row_begin=rows[:B], row_end=rows[B:])
            else:                                                                # qk observed on the unfused path
                for l, h, s in zip(self.sel_layer.tolist(), self.sel_head.tolist(), self.sel_slot.tolist()):
                    ring[:, s].copy_(captured[l][:, h])
            del q_out, k_out, captured
            self._mark(st, "qk_rows>")
            # host, while the GPU is busy with the forward pass: tokens -> words, descriptors, which log-probs to read
            n_gather = 0
            gather_rows, gather_toks = [], []
            for b, (job, toks) in enumerate(zip(st.jobs, fed)):
                end_token = tk.timestamp_begin + round(min(N_SAMPLES, job.n_crop_samples) // AUDIO_SAMPLES_PER_TOKEN)   # :1240
                utoks = toks[i_start:] + [end_token]
                st.tokens.append(utoks)
                unit = prepare_unit(utoks, None, tk, use_space=self.use_space,
                                    refine_whisper_precision_nframes=self.refine,
                                    remove_punctuation_from_words=self.remove_punct, detect_disfluencies=self.disfl,
                                    start_of_padding=None, qk_selected=ring[b, :, i_start - 1:len(toks)])
                st.units.append(unit)
                plan = []
                if unit is not None and self.want_conf:
                    # which (decoder position, token) pairs the reference reads (:1285-1292): position i_start + k for the
                    # k-th token of the running word list; inner timestamp tokens of a whole-window alignment are NOT
                    # stepped over (the reference's own bookkeeping, reproduced)
                    i_tok = i_start
                    for pieces, ids in planned_words(unit):
                        kept = ids
                        if self.incl_punct:                    # (sic) the reference strips when this is True (:1288-1291)
                            kept = ids[:len(strip_trailing_punctuation(pieces))]
                        plan.append((n_gather, len(kept)))
                        gather_rows.extend(range(b * T_max + i_tok, b * T_max + i_tok + len(kept)))
                        gather_toks.extend(kept)
                        n_gather += len(kept)
                        i_tok += len(pieces)
                st.plans.append(plan)
            # where each window's zero padding starts: queued right behind the log-mel, long since on the host
            pad_host = pad_copy.wait()
            batch = AlignmentBatch(workspace=self.workspace, extra_words=n_gather, stage_set=ss)
            for b, u in enumerate(st.units):
                if u is not None:
                    set_padding(u, None if int(pad_host[b]) < 0 else int(pad_host[b]))
                batch.add(u)
            self._mark(st, "align<")
            if ss is not None:
                ss.start_timeline(self.timeline is not None)
            batch.launch()
            self._mark(st, "align>")
            if ss is not None and ss.timeline:
                # the DTW ran on the set's other stream: its own event pair (start = its cost stage done)
                st.marks.extend(m for stage, ev0, ev1 in ss.timeline if stage == "dtw" for m in (("dtw<", ev0), ("dtw>", ev1)))
                ss.timeline = None                  # (the events live on in st.marks: not recycled)
            self._mark(st, "logprob<")
            if n_gather and batch.units:
                gt = self._to_device(np.concatenate([np.asarray(gather_rows, dtype=np.int32),
                                                      np.asarray(gather_toks, dtype=np.int32)]), st.keep)
                _lib.logprob_gather_rows(logits.reshape(B * T_max, -1), gt[:n_gather], gt[n_gather:],
                                         out=batch.extra.view(torch.float32))
            self._mark(st, "logprob>")
            batch.fetch()
            st.batch = batch
            st.keep.extend([ring, logits, tok_dev])
        return st

    # ------------------------------------------------------------------ host: the one wait of the sub-batch
    def collect(self, st: _Stage):
        words = iter(st.batch.collect())
        lp_all = None
        if st.batch.extra_words and st.batch.units:
            lp_all = torch.from_numpy(st.batch.extra_host.view(np.float32).copy())
        out = []
        for job, unit, plan, utoks, (first, last) in zip(st.jobs, st.units, st.plans, st.tokens, st.checks):
            ws = next(words) if unit is not None else []
            wl = None
            if self.want_conf:
                real = [w for w in ws if w["tokens"] or w["text"] != "[*]"]      # disfluency marks carry no tokens
                assert len(real) == len(plan), f"planned {len(plan)} words, aligned {len(real)}"
                it = iter(plan)
                wl = []
                for w in ws:
                    if w["tokens"] or w["text"] != "[*]":
                        off, n = next(it)
                        wl.append(lp_all[off:off + n] if n else lp_all[:0] if lp_all is not None else torch.empty(0))
                    else:
                        wl.append(torch.empty(0))
            out.append(WindowResult(ws, wl, utoks, first, last, job.tag))
        for k in st.keep:
            if isinstance(k, _lib.PinnedUpload):
                k.release()
        st.keep.clear()
        if self.timeline is not None and st.marks:
            ev = dict(st.marks)
            row = {name[:-1]: ev[name].elapsed_time(ev[name[:-1] + ">"]) for name in ev if name.endswith("<")}
            row["span"] = st.marks[0][1].elapsed_time(st.marks[-1][1])      # first launch -> last kernel, idle gaps included
            self.timeline.append(row)
        return out


def align_windows(aligner: BatchedAligner, jobs, max_windows=MAX_WINDOWS_PER_LAUNCH):
    """Yield one WindowResult per job, in order.  Sub-batches of ``max_windows`` are pipelined: the next one's GPU work is
    queued before this one's words are assembled on the host."""
    jobs = list(jobs)
    prev = None
    for lo in range(0, len(jobs), max_windows):
        cur = aligner.launch(jobs[lo:lo + max_windows])
        if prev is not None:
            yield from aligner.collect(prev)
        prev = cur
    if prev is not None:
        yield from aligner.collect(prev)
