"""Word alignment of Whisper segments on the MI355X: host mirror of the
reference's ``perform_word_alignment`` seam
(/root/reference/whisper_timestamped/transcribe.py:1428-1793) on top of the
HIP kernels in libwtalign.so.

Split of work
  host (Python, integer/string logic) : frame window, token->word grouping,
      "too much text" truncation (:1516-1535), jumps -> word times.
  device (HIP, one batched launch set for any number of units): head select,
      median filter, softmax, head mean, column norm, padding mask, DTW,
      backtrack, jumps (:1540-1581, :1648-1652).

``perform_word_alignment`` keeps the reference's signature and return value
(one unit).  ``AlignmentBatch`` is the MI355X-first entry: queue any number of
units (segments of many 30 s windows), run them in one go.
"""
from __future__ import annotations

import logging
from dataclasses import dataclass

import threading

import numpy as np
import torch

from . import _lib
from .words import (AUDIO_TIME_PER_TOKEN, _punctuation, frame_window, split_tokens_on_spaces,
                    split_tokens_on_unicode, trailing_punctuation_counts, words_from_jumps)

logger = logging.getLogger("whisper_timestamped")


def head_pairs(alignment_heads):
    """None | sparse COO (L,H) bool tensor (what the reference carries) | (n,2) array -> list of (layer, head) or None."""
    if alignment_heads is None:
        return None
    if isinstance(alignment_heads, torch.Tensor):
        if alignment_heads.is_sparse:
            return [(int(l), int(h)) for l, h in alignment_heads.coalesce().indices().T.tolist()]
        if alignment_heads.dtype == torch.bool:
            return [(int(l), int(h)) for l, h in alignment_heads.nonzero().tolist()]
    return [(int(l), int(h)) for l, h in np.asarray(alignment_heads).reshape(-1, 2).tolist()]


def max_duration_from_padding(start_of_padding):
    """transcribe.py:1554-1558: find_start_padding(mfcc) // 2 (None stays None)."""
    if start_of_padding is None or start_of_padding < 0:
        return None
    return int(start_of_padding) // 2


@dataclass
class AlignmentUnit:
    tokens: list
    qk: torch.Tensor            # (n_sel, T, n_ctx) logits of the selected heads, on the GPU
    start_token: int
    end_token: int
    pad_from: int               # -1 = no mask
    words: list
    word_pieces: list
    word_ids: list
    punct_counts: list
    refine_nframes: int
    unfinished_decoding: bool
    detect_disfluencies: bool
    tokenizer: object
    tag: object = None          # caller's handle
    mel: object = None          # the window's log-mel, kept for the debug figure only (plot_word_alignment)

    @property
    def T(self):
        return len(self.tokens)

    @property
    def F(self):
        return self.end_token - self.start_token


def _gather_heads(attention_weights, pairs, device):
    """list of L tensors (1,H,T,n_ctx) -> (A,T,n_ctx) on `device` (only the selected heads move)."""
    if pairs is None:
        ws = [torch.as_tensor(w).to(device) for w in attention_weights]
        cat = torch.cat(ws)                                   # (L,H,T,n_ctx), transcribe.py:1512
        return cat.reshape(-1, *cat.shape[-2:]).contiguous()
    rows = [torch.as_tensor(attention_weights[l])[0, h].to(device) for l, h in pairs]
    return torch.stack(rows).contiguous()


def prepare_unit(tokens, attention_weights, tokenizer, use_space=True, mfcc=None, refine_whisper_precision_nframes=0,
                 remove_punctuation_from_words=False, include_punctuation_in_timing=False, unfinished_decoding=False,
                 alignment_heads=None, detect_disfluencies=True, start_of_padding="auto", device=None, tag=None,
                 qk_selected=None):
    """Host part of perform_word_alignment up to the kernel call.  Returns an
    AlignmentUnit, or None for the empty segment of transcribe.py:1478-1481.

    ``attention_weights`` is the reference's list of per-layer (1,H,T,n_ctx)
    tensors; alternatively ``qk_selected`` is an already head-selected
    (A_sel, T, n_ctx) GPU tensor (e.g. a QKCaptureRing view) and
    ``attention_weights`` / ``alignment_heads`` are ignored."""
    tokens = [int(t) for t in tokens]
    show = lambda: tokenizer.decode_with_timestamps(tokens)  # noqa: E731
    win = frame_window(tokens, tokenizer.timestamp_begin, refine_whisper_precision_nframes, show)
    if win is None:
        return None
    start_token, end_token = win

    if qk_selected is not None:
        assert qk_selected.shape[-2] == len(tokens), \
            f"Attention weights have wrong shape: {qk_selected.shape[-2]} (expected {len(tokens)})."
    else:
        for w in attention_weights:
            assert w.shape[-2] == len(tokens), f"Attention weights have wrong shape: {w.shape[-2]} (expected {len(tokens)})."
    num_frames = end_token - start_token
    if len(tokens) > num_frames:                               # transcribe.py:1516-1535
        logger.warning(f"Too much text ({len(tokens)} tokens) for the given number of frames ({num_frames}) in: "
                       f"{show()}\nThe end of the text will be removed.")
        keep = num_frames - 1
        if qk_selected is not None:
            qk_selected = torch.cat([qk_selected[:, :keep], qk_selected[:, -1:]], dim=1)
        else:
            attention_weights = [torch.cat([torch.as_tensor(w)[:, :, :keep, :], torch.as_tensor(w)[:, :, -1:, :]], dim=-2)
                                 for w in attention_weights]
        return prepare_unit(
            tokens[:keep] + [tokens[-1]], attention_weights,
            tokenizer, use_space=use_space, mfcc=mfcc,
            refine_whisper_precision_nframes=refine_whisper_precision_nframes,
            remove_punctuation_from_words=remove_punctuation_from_words,
            include_punctuation_in_timing=False,               # the reference's recursion drops this argument
            unfinished_decoding=True, alignment_heads=alignment_heads, detect_disfluencies=detect_disfluencies,
            start_of_padding=start_of_padding, device=device, tag=tag, qk_selected=qk_selected)

    splitter = split_tokens_on_spaces if use_space else split_tokens_on_unicode
    words, word_pieces, word_ids = splitter(tokens, tokenizer, remove_punctuation_from_words=remove_punctuation_from_words)
    punct_counts = trailing_punctuation_counts(word_pieces, include_punctuation_in_timing)

    if qk_selected is not None:
        _lib._need_cuda(qk_selected, "qk_selected")
        qk, device = qk_selected, qk_selected.device
    else:
        if device is None:
            device = next((w.device for w in attention_weights if isinstance(w, torch.Tensor) and w.is_cuda),
                          torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None)
        if device is None:
            raise _lib.WtError("no GPU: the word-alignment kernels have no CPU fallback")
        qk = _gather_heads(attention_weights, head_pairs(alignment_heads), device)
    assert end_token <= qk.shape[-1]

    if start_of_padding == "auto":
        start_of_padding = None
        if mfcc is not None:
            sp = int(_lib.find_start_padding(torch.as_tensor(mfcc).to(device).float().reshape(1, *mfcc.shape[-2:]))[0])
            start_of_padding = None if sp < 0 else sp
    unit = AlignmentUnit(tokens, qk, start_token, end_token, -1, words, word_pieces, word_ids, punct_counts,
                         refine_whisper_precision_nframes, unfinished_decoding, detect_disfluencies, tokenizer, tag)
    return set_padding(unit, start_of_padding)


def set_padding(unit: AlignmentUnit, start_of_padding):
    """transcribe.py:1554-1565: where the zero padding of the window's log-mel starts (find_start_padding; None or a
    negative value = no padding) -> the unit's pad mask.  Separate from prepare_unit so that a batch can split its
    tokens into words while the GPU is still computing the log-mel the padding is read from."""
    max_duration = max_duration_from_padding(start_of_padding)
    unit.pad_from = -1
    if max_duration:
        if unit.start_token >= max_duration:
            logger.warning("Got start time outside of audio boundary")
        else:
            unit.pad_from = max_duration
    return unit


class _Slot:
    """Reusable buffers of one in-flight batch: pinned descriptor staging, device descriptors, cost matrices, and ONE
    int32 result record (jumps | moved starts | caller's extra words) with its pinned host mirror."""

    def __init__(self, device):
        self.device = device
        self.descs_host = self.descs_dev = self.cost = self.result = self.result_host = None
        self.event = torch.cuda.Event()
        self.head_idx = {}

    @staticmethod
    def _grown(n):
        return max(int(n * 1.5), 1024)

    def reserve(self, desc_bytes, n_cost, n_result):
        if self.descs_host is None or self.descs_host.numel() < desc_bytes:
            n = self._grown(desc_bytes)
            self.descs_host = torch.empty(n, dtype=torch.uint8).pin_memory()
            self.descs_dev = torch.empty(n, dtype=torch.uint8, device=self.device)
        if self.cost is None or self.cost.numel() < n_cost:
            self.cost = torch.empty(self._grown(n_cost), dtype=torch.float32, device=self.device)
        if self.result is None or self.result.numel() < n_result:
            n = self._grown(n_result)
            self.result = torch.empty(n, dtype=torch.int32, device=self.device)
            self.result_host = torch.empty(n, dtype=torch.int32).pin_memory()

    def heads(self, n_sel):
        if n_sel not in self.head_idx:
            self.head_idx[n_sel] = torch.arange(n_sel, dtype=torch.int32, device=self.device)
        return self.head_idx[n_sel]


class Workspace:
    """Pool of _Slots of one device.  A transcription session keeps one: after the first windows nothing is allocated
    per batch any more (descriptor staging, cost, result record and its pinned mirror are all reused), and the only
    host<->device synchronisation of a batch is the wait on its result event in ``AlignmentBatch.collect``."""

    def __init__(self, device):
        self.device = torch.device(device)
        self._free = []
        self._lock = threading.Lock()

    def acquire(self):
        with self._lock:
            if self._free:
                return self._free.pop()
        return _Slot(self.device)

    def release(self, slot):
        with self._lock:
            self._free.append(slot)


_WORKSPACES = {}
_WORKSPACES_LOCK = threading.Lock()


def default_workspace(device):
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    with _WORKSPACES_LOCK:
        if device not in _WORKSPACES:
            _WORKSPACES[device] = Workspace(device)
        return _WORKSPACES[device]


class AlignmentBatch:
    """Queue AlignmentUnits, run cost + DTW for all of them with one launch set, then turn jumps into words on the
    host.  ``run()`` = ``launch()`` (asynchronous: descriptors up, kernels, nothing waits) + ``collect()`` (ONE wait on
    the batch's result event, then host-side word assembly).  Between the two the caller may queue more GPU work
    (the next tokens of the decoder, another batch); ``extra_words`` reserves room at the end of the result record
    for values of its own (the batched naive strategy puts its chosen-token log-probabilities there) so that one
    copy brings everything to the host."""

    def __init__(self, medfilt_width=9, qk_scale=1.0, keep_cost=False, want_path=False, workspace=None, extra_words=0,
                 subwords_can_be_empty=True, stage_set=None, plot=False):
        self.units: list[AlignmentUnit] = []
        # plot_word_alignment (True: show, "<prefix>": save): one figure per unit once its words exist -- collect() then
        # reads the units' cost matrices and warping paths back (plotting.py); nothing changes for plot=False
        self.plot = plot
        self._release_after_figures = bool(plot) and not keep_cost     # (a caller that asked for the matrices keeps them)
        if plot:
            keep_cost = want_path = True
        self.medfilt_width, self.qk_scale = medfilt_width, qk_scale
        # transcribe.py:1571-1580: symmetric1, or the pattern without the previous-token/same-frame move
        self.step_pattern = _lib.WT_STEP_SYMMETRIC1 if subwords_can_be_empty else _lib.WT_STEP_NO_EMPTY_SUBWORDS
        self.keep_cost, self.want_path = keep_cost, want_path
        self.cost = self.jumps = self.descs = None
        self.path_i = self.path_j = self.path_len = self.dist = None
        self.workspace, self.extra_words = workspace, int(extra_words)
        self.extra = None               # device int32 view of the caller's words (valid after launch())
        self._slot = self._order = None
        self._launched = self._fetched = False
        # pipeline.StageSet: the cost stage on its low-priority stream (which must be the caller's CURRENT stream), the DTW
        # on its high-priority one behind it by event -- for batches without small units (those take the fused tail kernel)
        self.stage_set = stage_set
        self._dtw_elsewhere = False

    def add(self, unit: AlignmentUnit | None):
        if unit is not None:
            assert not self._launched, "the batch is already in flight"
            self.units.append(unit)
        return unit

    def launch(self):
        """Asynchronous half: after this call the units' QK rows may be overwritten by later work ON THE SAME STREAM."""
        assert not self._launched
        self._launched = True
        if not self.units:
            return self
        # the library wants units grouped by F class; results are handed back in the caller's order
        order = _lib.launch_order([(u.T, u.F) for u in self.units])
        units = [self.units[i] for i in order]
        dev = units[0].qk.device
        dt = units[0].qk.dtype
        esz = units[0].qk.element_size()
        base = min(u.qk.data_ptr() for u in units)
        ws = self.workspace or default_workspace(dev)
        assert ws.device == dev, f"workspace of {ws.device} used for units on {dev}"
        slot = ws.acquire()
        try:
            self._launch_on(slot, ws, units, order, dev, dt, esz, base)
        except BaseException:
            # A refused batch (WT_E_UNSUPPORTED ...) must not cost the pool a slot -- but an entry point may have queued
            # kernels on this slot's buffers before a LATER one refused (align ok, disfluency refused): the slot only
            # goes back once the stream has drained, so nobody else can be handed buffers that are still being written.
            drained = True
            if dev.type == "cuda":
                try:
                    torch.cuda.current_stream(dev).synchronize()
                except Exception:                  # noqa: BLE001 -- a dead device: the slot is dropped, not pooled
                    drained = False
            if drained:
                ws.release(slot)
            raise
        return self

    def _launch_on(self, slot, ws, units, order, dev, dt, esz, base):
        n_units = len(units)
        n_sel = units[0].qk.shape[0]
        desc_bytes = n_units * _lib.SEG_DTYPE.itemsize
        slot.reserve(desc_bytes, 0, 0)
        descs = slot.descs_host.numpy()[:desc_bytes].view(_lib.SEG_DTYPE)      # written in place in pinned memory
        descs[:] = 0
        for d, u in zip(descs, units):
            assert u.qk.dtype == dt and u.qk.device == dev and u.qk.shape[0] == n_sel and u.qk.stride(2) == 1
            off = u.qk.data_ptr() - base
            assert off % esz == 0
            d["qk_offset"] = off // esz
            d["head_stride"] = u.qk.stride(0)
            d["row_stride"] = u.qk.stride(1)
            d["T"], d["F"] = u.T, u.F
            d["start_token"], d["pad_from"] = u.start_token, u.pad_from
        n_cost, n_jumps, n_path = _lib.layout_outputs(descs)
        disfl = any(u.detect_disfluencies for u in units)
        n_result = n_jumps * (2 if disfl else 1) + self.extra_words
        slot.reserve(desc_bytes, n_cost, n_result)
        descs = slot.descs_host.numpy()[:desc_bytes].view(_lib.SEG_DTYPE)      # (reserve never reallocates the staging)
        with _lib.on_device(dev) as stream:
            descs_dev = slot.descs_dev[:desc_bytes]
            descs_dev.copy_(slot.descs_host[:desc_bytes], non_blocking=True)
            cost = slot.cost[:n_cost]
            jumps = slot.result[:n_jumps]
            if self.want_path:
                self.path_i = torch.empty(n_path, dtype=torch.int32, device=dev)
                self.path_j = torch.empty(n_path, dtype=torch.int32, device=dev)
                self.path_len = torch.empty(n_units, dtype=torch.int32, device=dev)
                self.dist = torch.empty(n_units, dtype=torch.float64, device=dev)
            L = _lib.load()
            ss = self.stage_set
            if ss is not None and ss.hi is not ss.lo and not any(_lib.small_unit(u.T, u.F) for u in units):
                assert ss.lo.cuda_stream == stream, "AlignmentBatch(stage_set=): launch() with the set's low-priority stream current"
                dt_code = {torch.float32: 0, torch.float16: 1}[dt]
                heads = slot.heads(n_sel).data_ptr()
                ss.run("cost", lambda st_: _lib._check(L.wt_cost_batch(
                    base, dt_code, descs.ctypes.data, descs_dev.data_ptr(), n_units, heads, n_sel, self.medfilt_width,
                    float(self.qk_scale), cost.data_ptr(), st_), "wt_cost_batch"))
                ss.run("dtw", lambda st_: _lib._check(L.wt_dtw_batch_pattern(
                    cost.data_ptr(), descs.ctypes.data, descs_dev.data_ptr(), n_units, self.step_pattern, jumps.data_ptr(),
                    _lib._ptr(self.path_i), _lib._ptr(self.path_j), _lib._ptr(self.path_len), _lib._ptr(self.dist), st_),
                    "wt_dtw_batch_pattern"), after=("cost",))
                self._dtw_elsewhere = True
                if disfl:
                    ss.wait_for(ss.lo, ("dtw",))
            elif self.step_pattern == _lib.WT_STEP_SYMMETRIC1:
                # units of the reference's per-segment shape take the fused small-unit kernel; their cost matrices only
                # go to HBM when somebody reads them afterwards (keep_cost, the disfluency kernel)
                flags = (_lib.WT_ALIGN_KEEP_COST if (self.keep_cost or disfl) else 0) | \
                    (0 if FUSED_SMALL_UNITS else _lib.WT_ALIGN_NO_FUSED_SMALL_UNITS)
                rc = L.wt_align_batch_v3(base, {torch.float32: 0, torch.float16: 1}[dt], descs.ctypes.data, descs_dev.data_ptr(),
                                         n_units, slot.heads(n_sel).data_ptr(), n_sel, self.medfilt_width, float(self.qk_scale),
                                         cost.data_ptr(), jumps.data_ptr(), _lib._ptr(self.path_i), _lib._ptr(self.path_j),
                                         _lib._ptr(self.path_len), _lib._ptr(self.dist), flags, stream)
                _lib._check(rc, "wt_align_batch_v3")
            else:
                rc = L.wt_cost_batch(base, {torch.float32: 0, torch.float16: 1}[dt], descs.ctypes.data, descs_dev.data_ptr(),
                                     n_units, slot.heads(n_sel).data_ptr(), n_sel, self.medfilt_width, float(self.qk_scale),
                                     cost.data_ptr(), stream)
                _lib._check(rc, "wt_cost_batch")
                rc = L.wt_dtw_batch_pattern(cost.data_ptr(), descs.ctypes.data, descs_dev.data_ptr(), n_units,
                                            self.step_pattern, jumps.data_ptr(), _lib._ptr(self.path_i),
                                            _lib._ptr(self.path_j), _lib._ptr(self.path_len), _lib._ptr(self.dist), stream)
                _lib._check(rc, "wt_dtw_batch_pattern")
            if disfl:                                          # token starts moved to their last attention peak
                rc = L.wt_disfluency_batch(cost.data_ptr(), descs_dev.data_ptr(), n_units, jumps.data_ptr(),
                                           slot.result[n_jumps:2 * n_jumps].data_ptr(), DISFLUENCY_MIN_PROMINENCE,
                                           DISFLUENCY_MIN_WIDTH, stream)
                _lib._check(rc, "wt_disfluency_batch")
        self._slot, self._order, self._sorted_units = slot, order, units
        self._n_jumps, self._disfl, self._n_result = n_jumps, disfl, n_result
        self.descs, self.cost, self.jumps = descs.copy(), cost, jumps
        self.extra = slot.result[n_result - self.extra_words:n_result] if self.extra_words else None

    def release(self):
        """Hand the batch's buffers back to the workspace (keep_cost=True batches hold them until told: the cost
        matrices live there).  Also the exit of ``with AlignmentBatch(keep_cost=True) as batch:``."""
        if self._slot is not None:
            (self.workspace or default_workspace(self._slot.device)).release(self._slot)
            self._slot = None
            self.cost = self.jumps = self.extra = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.release()
        return False

    def fetch(self):
        """Queue the ONE device->host copy of the result record (KBs) behind whatever has been launched so far."""
        if not self._launched:
            self.launch()
        if self._fetched or not self.units:
            return self
        self._fetched = True
        slot = self._slot
        with torch.cuda.device(slot.device):
            if self._dtw_elsewhere:                  # the jumps come from the set's other stream
                self.stage_set.wait_for(torch.cuda.current_stream(slot.device), ("dtw",))
            slot.result_host[:self._n_result].copy_(slot.result[:self._n_result], non_blocking=True)
            slot.event.record(torch.cuda.current_stream(slot.device))
        return self

    def collect(self):
        """Wait for the record (the one synchronisation of the batch) and assemble the words, caller's order."""
        if not self.units:
            return []
        self.fetch()
        slot, units, order = self._slot, self._sorted_units, self._order
        slot.event.synchronize()
        host = slot.result_host[:self._n_result].numpy().copy()
        n_jumps = self._n_jumps
        jumps_host = host[:n_jumps]
        starts_host = host[n_jumps:2 * n_jumps] if self._disfl else None
        self.extra_host = host[self._n_result - self.extra_words:] if self.extra_words else None
        out = [None] * len(units)
        self._slot_of = [0] * len(units)             # caller's unit index -> descriptor index
        for k, (d, u) in enumerate(zip(self.descs, units)):
            j0 = int(d["jumps_offset"])
            jm = jumps_host[j0:j0 + u.T + 1].astype(np.int64)
            js = starts_host[j0:j0 + u.T + 1].astype(np.int64) if starts_host is not None else None
            out[order[k]] = finish_unit(u, jm, js)
            self._slot_of[order[k]] = k
        if self.plot:
            self._draw(out, jumps_host)
            if self._release_after_figures:
                self.release()
        if not self.keep_cost:                       # the cost matrices live in the slot: hand it back unless asked to keep
            self.release()
        return out

    def _draw(self, out, jumps_host):
        """plot_word_alignment: one figure per unit, in the caller's order (transcribe.py:1586-1646, 1685-1700, 1756-1781)."""
        from . import plotting
        for i, unit in enumerate(self.units):
            d = self.descs[self._slot_of[i]]
            j0 = int(d["jumps_offset"])
            path_tokens, path_frames = self.unit_path(i)
            plotting.alignment_figure(unit, self.unit_cost(i).cpu().numpy(), path_tokens.cpu().numpy(), path_frames.cpu().numpy(),
                                      jumps_host[j0:j0 + unit.T + 1], out[i], self.plot)

    def run(self):
        return self.launch().collect()

    def unit_cost(self, k):
        assert self.keep_cost, "AlignmentBatch(keep_cost=True) keeps the cost matrices after collect()"
        k = self._slot_of[k]
        d = self.descs[k]
        c0 = int(d["cost_offset"])
        return self.cost[c0:c0 + int(d["T"]) * int(d["F"])].reshape(int(d["T"]), int(d["F"]))

    def unit_path(self, k):
        k = self._slot_of[k]
        d = self.descs[k]
        n = int(self.path_len[k])
        p0 = int(d["path_offset"])
        return self.path_i[p0:p0 + n], self.path_j[p0:p0 + n]


FUSED_SMALL_UNITS = True      # False: every unit through the batched kernels (measurements, tests)
DISFLUENCY_MIN_PROMINENCE, DISFLUENCY_MIN_WIDTH = 0.02, 3.0      # find_peaks arguments of transcribe.py:1663-1666


def detect_disfluences(unit: AlignmentUnit, jumps, jumps_start):
    """transcribe.py:1656-1683.  The numeric half -- scipy.signal.find_peaks on every token's span of the cost matrix,
    and where the last peak starts -- ran on the GPU (wt_disfluency_batch -> jumps_start); what is left for the host
    is the tokenizer's business: a moved start marks the token as a disfluency unless the token is punctuation, in
    which case the NEXT token inherits the whole span."""
    disfluences = {}
    for i_token in np.nonzero(jumps_start[:-1] != jumps[:-1])[0]:
        i_token = int(i_token)
        begin, end = int(jumps[i_token]), int(jumps[i_token + 1])
        if unit.tokenizer.decode_with_timestamps([unit.tokens[i_token]]) not in _punctuation:
            disfluences[i_token] = (begin, int(jumps_start[i_token]))
        else:
            disfluences[i_token + 1] = (begin, end)
    return jumps_start, disfluences


def planned_words(unit: AlignmentUnit, with_text: bool = False):
    """(pieces, ids) -- or (text, pieces, ids) -- of the words ``finish_unit`` will return for this unit, in order, known
    BEFORE the kernels run: which words exist is decided by the token split alone (words.words_from_jumps drops the
    timestamp words and "<|...|>" texts; a disfluency mark "[*]" may be inserted later but carries no tokens)."""
    keep = slice(1, None) if unit.unfinished_decoding else slice(1, -1)
    rows = [(text, pieces, ids) for text, pieces, ids in zip(unit.words[keep], unit.word_pieces[keep], unit.word_ids[keep])
            if not text.startswith("<|")]
    return rows if with_text else [(p, i) for _, p, i in rows]


def finish_unit(unit: AlignmentUnit, jumps, jumps_start=None):
    disfluences = None
    if unit.detect_disfluencies:
        assert jumps_start is not None, "detect_disfluencies: the batch must have run wt_disfluency_batch"
        jumps_start, disfluences = detect_disfluences(unit, jumps, jumps_start)
    else:
        jumps_start = jumps
    return words_from_jumps(jumps, jumps_start, unit.words, unit.word_pieces, unit.word_ids, unit.punct_counts,
                            unit.start_token * AUDIO_TIME_PER_TOKEN, unit.refine_nframes, unit.unfinished_decoding,
                            disfluences)


def perform_word_alignment(tokens, attention_weights, tokenizer, use_space=True, mfcc=None,
                           refine_whisper_precision_nframes=0, remove_punctuation_from_words=False,
                           include_punctuation_in_timing=False, unfinished_decoding=False, alignment_heads=None,
                           medfilt_width=9, qk_scale=1.0, detect_disfluencies=True, subwords_can_be_empty=True,
                           plot=False, debug=False):
    """Drop-in for the reference function (same arguments, same list of
    dict(text, start, end, tokens, tokens_indices)); the numerics run on the GPU."""
    unit = prepare_unit(tokens, attention_weights, tokenizer, use_space=use_space, mfcc=mfcc,
                        refine_whisper_precision_nframes=refine_whisper_precision_nframes,
                        remove_punctuation_from_words=remove_punctuation_from_words,
                        include_punctuation_in_timing=include_punctuation_in_timing,
                        unfinished_decoding=unfinished_decoding, alignment_heads=alignment_heads,
                        detect_disfluencies=detect_disfluencies)
    if unit is None:
        if debug:
            logger.debug(f"Got empty segment in {tokenizer.decode_with_timestamps(list(tokens))}")
        return []
    batch = AlignmentBatch(medfilt_width=medfilt_width, qk_scale=qk_scale, subwords_can_be_empty=subwords_can_be_empty, plot=plot)
    if plot:
        unit.mel = mfcc
    batch.add(unit)
    return batch.run()[0]
