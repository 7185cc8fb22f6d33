"""oracle/align_ref.py -- TEST INFRASTRUCTURE ONLY (the CPU oracle).

CPU restatement of the reference's word-alignment hot path.  Every function
cites the reference lines it follows (paths are into /root/reference/;
``transcribe.py`` = ``whisper_timestamped/transcribe.py``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module -- as the checker, never as the product path.

Wherever the reference calls a library that IS installed here (torch CPU ops,
``scipy.ndimage.median_filter``, ``scipy.signal.find_peaks``) the oracle makes the
identical call.  The DTW core (dtw-python, absent) is restated in C
(``oracle/dtw_ref.c``).

PARITY STATUS
* cost construction / jumps / word times / padding / confidence: pinned against
  the reference's own code, imported in the build container with stub modules
  for the absent third-party packages (``tests/golden/make_golden.py`` ->
  ``tests/golden/*.npz|json``); the stub ``dtw.dtw`` is backed by
  ``oracle/dtw_ref.c``.
* DTW core: "parity unpinned" against dtw-python itself; pinned against an
  exhaustive path enumeration and transformers' independent DTW on tie-free
  inputs (tests/test_oracle.py).
* word splitting: pinned by the reference's known-answer test
  tests/test_transcribe.py:722-902 (tests/golden/split_tokens_kat.json).
* the strategies around the alignment (hook state machine, naive re-run, confidences, post-fixers, VAD
  back-conversion) are NOT restated here: their parity is established directly against the output of the
  reference's own transcribe_timestamped (tests/golden/make_golden_transcribe.py ->
  tests/golden/transcribe_cases.json), with this module standing in for the kernels in the CPU run
  (tests/cpu_kernel_standin.py).
"""
from __future__ import annotations

import ctypes
import os
import string
import subprocess

import numpy as np
import torch
from scipy.ndimage import median_filter
from scipy.signal import find_peaks

# whisper.audio constants mirrored at transcribe.py:44-47
N_FRAMES = 3000
HOP_LENGTH = 160
SAMPLE_RATE = 16000
N_FFT = 400
AUDIO_TIME_PER_TOKEN = (HOP_LENGTH * 2) / SAMPLE_RATE  # 0.02 s
N_AUDIO_CTX = N_FRAMES // 2  # 1500 frames of 20 ms

# transcribe.py:1813
PUNCTUATION = "".join(c for c in string.punctuation if c not in ["-", "'"]) + "。，！？：”、…"
DISFLUENCY_MARK = "[*]"  # transcribe.py:70

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build_native(force: bool = False) -> str:
    """Compile oracle/dtw_ref.c -> oracle/liboracle.so (gcc)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "dtw_ref.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build_native()
        L = ctypes.CDLL(so)
        dp = ctypes.POINTER(ctypes.c_double)
        ip = ctypes.POINTER(ctypes.c_int32)
        L.wt_ref_dtw_cm.argtypes = [dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp, ip]
        L.wt_ref_dtw_cm.restype = ctypes.c_int
        L.wt_ref_backtrack.argtypes = [ip, ctypes.c_int, ctypes.c_int, ip, ip]
        L.wt_ref_backtrack.restype = ctypes.c_int
        L.wt_ref_dtw_jumps.argtypes = [dp, ctypes.c_int, ctypes.c_int, ip, ip, ip, ip, dp]
        L.wt_ref_dtw_jumps.restype = ctypes.c_int
        L.wt_ref_dtw_bruteforce_cost.argtypes = [dp, ctypes.c_int, ctypes.c_int]
        L.wt_ref_dtw_bruteforce_cost.restype = ctypes.c_double
        _LIB = L
    return _LIB


# --------------------------------------------------------------------------
# DTW (dtw-python semantics; see oracle/dtw_ref.c)
# --------------------------------------------------------------------------
class DtwResult:
    """Mimics the attributes of dtw-python's result that the reference reads
    (transcribe.py:1598,1648-1652): index1s/index2s (+ index1/index2, distance,
    costMatrix / directionMatrix when keep_internals)."""

    def __init__(self, index1, index2, distance, cm=None, sm=None):
        self.index1 = self.index1s = index1
        self.index2 = self.index2s = index2
        self.distance = distance
        self.costMatrix = cm
        self.directionMatrix = sm
        self.N, self.M = (cm.shape if cm is not None else (int(index1[-1]) + 1, int(index2[-1]) + 1))


def dtw_ref(cost: np.ndarray, step_pattern: int = 0, keep_internals: bool = False) -> DtwResult:
    """``dtw.dtw(cost, step_pattern=symmetric1)`` -- transcribe.py:1572,1581."""
    lm = np.ascontiguousarray(cost, dtype=np.float64)
    if lm.ndim != 2:
        raise ValueError("local cost must be 2-D")
    T, F = lm.shape
    cm = np.empty((T, F), dtype=np.float64)
    sm = np.empty((T, F), dtype=np.int32)
    L = _lib()
    dp = ctypes.POINTER(ctypes.c_double)
    ip = ctypes.POINTER(ctypes.c_int32)
    rc = L.wt_ref_dtw_cm(lm.ctypes.data_as(dp), T, F, step_pattern, cm.ctypes.data_as(dp), sm.ctypes.data_as(ip))
    if rc == -4:
        raise ValueError("NaN in local cost matrix")  # dtw-python: _error on NaN
    if rc != 0:
        raise RuntimeError(f"wt_ref_dtw_cm rc={rc}")
    i1 = np.empty(T + F, dtype=np.int32)
    i2 = np.empty(T + F, dtype=np.int32)
    n = L.wt_ref_backtrack(sm.ctypes.data_as(ip), T, F, i1.ctypes.data_as(ip), i2.ctypes.data_as(ip))
    if n < 0:
        raise ValueError(f"No warping path found compatible with the local constraints (rc={n})")
    return DtwResult(i1[:n].copy(), i2[:n].copy(), float(cm[-1, -1]), cm if keep_internals else None,
                     sm if keep_internals else None)


def dtw_bruteforce_cost(cost: np.ndarray) -> float:
    lm = np.ascontiguousarray(cost, dtype=np.float64)
    return float(_lib().wt_ref_dtw_bruteforce_cost(lm.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), *lm.shape))


def jumps_from_path(index1s: np.ndarray, index2s: np.ndarray) -> np.ndarray:
    """transcribe.py:1648-1652 (identical numpy calls)."""
    jumps = np.diff(index1s)
    jumps = np.pad(jumps, (1, 0), constant_values=1)
    jumps = jumps.astype(bool)
    jumps = index2s[jumps]
    jumps = np.pad(jumps, (0, 1), constant_values=index2s[-1])
    return jumps


# --------------------------------------------------------------------------
# Cost construction (transcribe.py:1540-1568)
# --------------------------------------------------------------------------
def select_heads_ref(attention_weights, start_token, end_token, alignment_heads=None) -> torch.Tensor:
    """transcribe.py:1512,1540-1545: cat layers, slice frames, pick heads.

    attention_weights: list of L tensors (1,H,T,1500); alignment_heads: None or
    an (n,2) integer array of (layer, head) pairs in COO order (what
    ``alignment_heads.indices().T`` yields)."""
    weights = torch.cat([torch.as_tensor(w) for w in attention_weights])  # (L,H,T,1500)
    weights = weights[..., start_token:end_token].cpu()
    if alignment_heads is None:
        return weights.reshape(-1, *weights.shape[-2:])
    return torch.stack([weights[int(l)][int(h)] for l, h in alignment_heads])


def cost_matrix_ref(sel: torch.Tensor, medfilt_width: int = 9, qk_scale: float = 1.0,
                    max_duration=None, start_token: int = 0, return_f32: bool = False) -> np.ndarray:
    """(A,T,F) fp32 QK logits of the selected heads -> (T,F) f64 local cost.

    transcribe.py:1546-1550 (median -> softmax -> mean -> column L2 norm ->
    negate as double), 1560-1565 (padding mask with the reference's
    absolute-index-used-as-relative quirk), 1568 (cost[0,0] = min)."""
    w = median_filter(sel, (1, 1, medfilt_width))              # :1546 (scipy, mode='reflect')
    w = torch.tensor(w * qk_scale).softmax(dim=-1)             # :1547
    w = w.mean(axis=(0))                                       # :1548
    w = w / w.norm(dim=-2, keepdim=True)                       # :1549
    if return_f32:
        return w.numpy()
    w = -w.double().numpy()                                    # :1550
    if max_duration:                                           # :1561
        if start_token >= max_duration:                        # :1562 (warning only)
            pass
        else:
            w[:-1, max_duration:] = 0                          # :1565
    w[0, 0] = w.min()                                          # :1568
    return w


def find_start_padding_ref(mfcc: torch.Tensor):
    """transcribe.py:1795-1805, same control flow."""
    last_mfcc = mfcc[0, :, -1]
    if torch.min(last_mfcc) == torch.max(last_mfcc) == 0:
        candidate_index = mfcc.shape[-1] - 2
        while candidate_index > 0:
            candidate = mfcc[0, :, candidate_index]
            if not torch.equal(candidate, last_mfcc):
                return candidate_index + 1
            candidate_index -= 1
        return 0
    return None


def max_duration_ref(mfcc):
    """transcribe.py:1554-1558."""
    if mfcc is None:
        return None
    md = find_start_padding_ref(mfcc)
    if md is not None:
        md = md // 2
    return md


# --------------------------------------------------------------------------
# Frame window (transcribe.py:1466-1494)
# --------------------------------------------------------------------------
def frame_window_ref(tokens, timestamp_begin, refine_nframes=0):
    """Returns (start_token, end_token) in 20 ms frames, or None for the empty
    segment of transcribe.py:1478-1481.  Raises like the reference."""
    assert len(tokens) > 1
    start_token = tokens[0] - timestamp_begin
    end_token = tokens[-1] - timestamp_begin
    if start_token < 0:
        raise RuntimeError("Missing start token")                          # :1472
    if len(tokens) == 1 or end_token < 0:
        end_token = N_FRAMES // 2                                          # :1477
    if end_token == start_token and refine_nframes == 0:
        return None                                                        # :1481
    end_token = min(N_FRAMES // 2, max(end_token, start_token + len(tokens)))  # :1484
    if refine_nframes > 0:                                                 # :1487-1489
        start_token = max(start_token - refine_nframes, 0)
        end_token = min(end_token + refine_nframes, N_FRAMES // 2)
    if end_token <= start_token:
        raise RuntimeError(f"Got segment with null or negative duration: {start_token} {end_token}")  # :1492
    return start_token, end_token


# --------------------------------------------------------------------------
# Token -> word grouping (transcribe.py:1815-1868)
# --------------------------------------------------------------------------
def split_tokens_on_unicode_ref(tokens, tokenizer, remove_punctuation_from_words=False, isolate_punctuations=False):
    """transcribe.py:1815-1842."""
    words, word_tokens, word_tokens_indices = [], [], []
    pending = []
    for tok in tokens:
        pending.append(tok)
        text = tokenizer.decode_with_timestamps(
            [t for t in pending if t < tokenizer.eot or t >= tokenizer.timestamp_begin])
        if "\ufffd" in text:
            continue
        blanks = [""] * (len(pending) - 1)
        is_punct = (not isolate_punctuations) and bool(text.strip() and text.strip() in PUNCTUATION)
        after_special = len(word_tokens_indices) > 0 and word_tokens_indices[-1][-1] >= tokenizer.timestamp_begin
        if is_punct and not after_special:
            if not words:
                words, word_tokens = [""], [[]]
                # NB: the reference leaves word_tokens_indices empty here and
                # would raise IndexError two lines below (transcribe.py:1829-1835);
                # reproduced.
            if not remove_punctuation_from_words:
                words[-1] += text
            word_tokens[-1].extend(blanks + [text])
            word_tokens_indices[-1].extend(pending)
        else:
            words.append(text)
            word_tokens.append(blanks + [text])
            word_tokens_indices.append(pending)
        pending = []
    return words, word_tokens, word_tokens_indices


def split_tokens_on_spaces_ref(tokens, tokenizer, remove_punctuation_from_words=False):
    """transcribe.py:1845-1868."""
    subs, sub_toks, sub_idx = split_tokens_on_unicode_ref(
        tokens, tokenizer, remove_punctuation_from_words=remove_punctuation_from_words)
    words, word_tokens, word_tokens_indices = [], [], []
    n = len(subs)
    for i in range(n):
        sub = subs[i]
        special = sub_idx[i][0] >= tokenizer.timestamp_begin
        prev_special = i > 0 and sub_idx[i - 1][0] >= tokenizer.timestamp_begin
        next_special = i < n - 1 and sub_idx[i + 1][0] >= tokenizer.timestamp_begin
        prev_space = i > 0 and not subs[i - 1].strip()
        is_space = not sub.strip()
        with_space = sub.startswith(" ") and not is_space
        punct = (not is_space) and sub.strip() in PUNCTUATION
        if special or (not prev_space and (prev_special or (with_space and not punct) or (is_space and not next_special))):
            words.append(sub.strip())
            word_tokens.append(sub_toks[i])
            word_tokens_indices.append(sub_idx[i])
        else:
            words[-1] = words[-1] + sub.strip()
            word_tokens[-1].extend(sub_toks[i])
            word_tokens_indices[-1].extend(sub_idx[i])
    return words, word_tokens, word_tokens_indices


# --------------------------------------------------------------------------
# Full alignment of one segment (transcribe.py:1428-1793, plotting omitted)
# --------------------------------------------------------------------------
def jumps_start_ref(cost, jumps, min_prominence=0.02, min_width=3):
    """transcribe.py:1656-1672 without the tokenizer part: the (possibly moved) start frame of every token row.
    cost: (T,F) local-cost matrix (f64 or f32; scipy works in f64 either way), jumps: T+1 frames.  Uses
    scipy.signal.find_peaks itself, like the reference."""
    jumps = np.asarray(jumps)
    out = jumps.copy()
    for t, (begin, end) in enumerate(zip(jumps[:-1], jumps[1:])):
        peaks, props = find_peaks(-np.asarray(cost[t, begin:end], dtype=np.float64), width=min_width, prominence=min_prominence)
        if len(peaks) > 1:
            out[t] = round(props["left_ips"][-1]) + begin
    return out


def jumps_start_restated(cost, jumps, min_prominence=0.02, min_width=3.0):
    """The same answer WITHOUT scipy: its algorithm (_local_maxima_1d, _peak_prominences, _peak_widths of
    scipy/signal/_peak_finding_utils.pyx) spelled out loop by loop -- the text the HIP kernel wt_peaks.hip follows.
    tests/test_oracle.py holds it against jumps_start_ref on random profiles."""
    jumps = np.asarray(jumps)
    out = jumps.copy()
    for t, (begin, end) in enumerate(zip(jumps[:-1], jumps[1:])):
        x = -np.asarray(cost[t, begin:end], dtype=np.float64)
        n = len(x)
        kept, last_left = 0, 0.0
        i, i_last = 1, n - 1
        while i < i_last:
            if x[i - 1] < x[i]:
                ahead = i + 1
                while ahead < i_last and x[ahead] == x[i]:
                    ahead += 1
                if x[ahead] < x[i]:
                    peak = (i + ahead - 1) // 2
                    i = ahead
                    xp = x[peak]
                    left_min = right_min = xp
                    lb = rb = peak
                    k = peak
                    while k >= 0 and x[k] <= xp:
                        if x[k] < left_min:
                            left_min, lb = x[k], k
                        k -= 1
                    k = peak
                    while k <= i_last and x[k] <= xp:
                        if x[k] < right_min:
                            right_min, rb = x[k], k
                        k += 1
                    prominence = xp - max(left_min, right_min)
                    if min_prominence <= prominence:
                        height = xp - prominence * 0.5
                        k = peak
                        while lb < k and height < x[k]:
                            k -= 1
                        left_ip = float(k)
                        if x[k] < height:
                            left_ip += (height - x[k]) / (x[k + 1] - x[k])
                        k = peak
                        while k < rb and height < x[k]:
                            k += 1
                        right_ip = float(k)
                        if x[k] < height:
                            right_ip -= (height - x[k]) / (x[k - 1] - x[k])
                        if min_width <= right_ip - left_ip:
                            kept += 1
                            last_left = left_ip
            i += 1
        if kept > 1:
            out[t] = round(last_left) + begin
    return out


def perform_word_alignment_ref(tokens, attention_weights, tokenizer, use_space=True, mfcc=None,
                               refine_whisper_precision_nframes=0, remove_punctuation_from_words=False,
                               include_punctuation_in_timing=False, unfinished_decoding=False,
                               alignment_heads=None, medfilt_width=9, qk_scale=1.0,
                               detect_disfluencies=True, return_internals=False, subwords_can_be_empty=True):
    """Same contract as the reference function.  ``alignment_heads`` is None
    or an (n,2) array of (layer, head).  ``subwords_can_be_empty=False`` selects
    the step pattern of transcribe.py:1575-1580 (no caller of the reference passes it)."""
    tokens = list(tokens)
    win = frame_window_ref(tokens, tokenizer.timestamp_begin, refine_whisper_precision_nframes)
    if win is None:
        return []
    start_token, end_token = win
    start_time = start_token * AUDIO_TIME_PER_TOKEN                                   # :1494

    split = split_tokens_on_spaces_ref if use_space else split_tokens_on_unicode_ref   # :1497
    words, word_tokens, word_tokens_indices = split(
        tokens, tokenizer, remove_punctuation_from_words=remove_punctuation_from_words)

    n_punct = [0 if len(w) == 1 or w[-1] not in PUNCTUATION else 1 for w in word_tokens]  # :1503-1506
    if include_punctuation_in_timing:
        n_punct[:-2] = [0] * (len(n_punct) - 2)                                      # :1508

    for w in attention_weights:
        assert w.shape[-2] == len(tokens), f"Attention weights have wrong shape: {w.shape[-2]} (expected {len(tokens)})."
    num_tokens = len(tokens)
    num_frames = end_token - start_token
    if num_tokens > num_frames:                                                       # :1516-1535
        return perform_word_alignment_ref(
            tokens[:num_frames - 1] + [tokens[-1]],
            [torch.cat([torch.as_tensor(w)[:, :, :num_frames - 1, :], torch.as_tensor(w)[:, :, -1:, :]], dim=-2)
             for w in attention_weights],
            tokenizer, use_space=use_space,
            refine_whisper_precision_nframes=refine_whisper_precision_nframes,
            medfilt_width=medfilt_width, qk_scale=qk_scale, alignment_heads=alignment_heads, mfcc=mfcc,
            remove_punctuation_from_words=remove_punctuation_from_words,
            detect_disfluencies=detect_disfluencies, unfinished_decoding=True,
            return_internals=return_internals)

    sel = select_heads_ref(attention_weights, start_token, end_token, alignment_heads)
    cost = cost_matrix_ref(sel, medfilt_width, qk_scale, max_duration_ref(mfcc), start_token)
    ali = dtw_ref(cost, step_pattern=0 if subwords_can_be_empty else 1)               # :1571-1581
    jumps = jumps_from_path(ali.index1s, ali.index2s)                                 # :1648-1652

    jumps_start = jumps
    disfluences = {}
    if detect_disfluencies:                                                           # :1656-1683
        jumps_start = jumps.copy()
        for i_token, (tok, begin, end) in enumerate(zip(tokens, jumps[:-1], jumps[1:])):
            att = -cost[i_token, begin:end]
            peaks, props = find_peaks(att, width=3, prominence=0.02)
            if len(peaks) > 1:
                if "left_ips" in props:
                    left = [round(x) for x in props["left_ips"]]
                else:
                    left = props["left_bases"]
                new_begin = left[-1] + begin
                jumps_start[i_token] = new_begin
                if new_begin != begin:
                    is_punct = tokenizer.decode_with_timestamps([tok]) in PUNCTUATION
                    if not is_punct:
                        disfluences[i_token] = (begin, jumps_start[i_token])
                    else:
                        disfluences[i_token + 1] = (begin, end)

    wb = np.pad(np.cumsum([len(t) for t in word_tokens]), (1, 0))                     # :1711-1712
    begin_times = jumps_start[wb[:-1]] * AUDIO_TIME_PER_TOKEN                         # :1713,1716
    end_times = jumps[wb[1:] - n_punct] * AUDIO_TIME_PER_TOKEN                        # :1714,1717

    if detect_disfluencies:                                                           # :1719-1736
        to_add = []
        i_start = 0
        for i_word, toks in enumerate(word_tokens[:-1]):
            i_end = i_start + len(toks)
            if i_start in disfluences and i_word > 0:
                b, e = disfluences[i_start]
                to_add.append((i_word, b * AUDIO_TIME_PER_TOKEN, e * AUDIO_TIME_PER_TOKEN))
            i_start = i_end
        for i_word, b, e in to_add[::-1]:
            words.insert(i_word, DISFLUENCY_MARK)
            word_tokens.insert(i_word, [])
            word_tokens_indices.insert(i_word, [])
            begin_times = np.insert(begin_times, i_word, b)
            end_times = np.insert(end_times, i_word, e)

    if not refine_whisper_precision_nframes:                                          # :1739-1742
        begin_times[1] = begin_times[0]
        end_times[-2] = end_times[-1]
    sl = slice(1, None) if unfinished_decoding else slice(1, -1)                      # :1743-1754
    words, word_tokens, word_tokens_indices = words[sl], word_tokens[sl], word_tokens_indices[sl]
    begin_times, end_times = begin_times[sl], end_times[sl]

    out = [
        dict(text=w, start=round(b + start_time, 2), end=round(e + start_time, 2), tokens=t, tokens_indices=ti)
        for w, b, e, t, ti in zip(words, begin_times, end_times, word_tokens, word_tokens_indices)
        if not w.startswith("<|")
    ]                                                                                  # :1783-1793
    if return_internals:
        return out, dict(cost=cost, jumps=jumps, jumps_start=jumps_start, start_token=start_token,
                         end_token=end_token, index1s=ali.index1s, index2s=ali.index2s)
    return out


# --------------------------------------------------------------------------
# Confidence path
# --------------------------------------------------------------------------
def logprobs_ref(logits: torch.Tensor) -> torch.Tensor:
    """``F.log_softmax(logits, dim=-1)`` on fp32 -- transcribe.py:875 (efficient,
    after the logit filters wrote -inf in place) and :1245 (naive, no filters)."""
    return torch.nn.functional.log_softmax(torch.as_tensor(logits).float(), dim=-1)


def token_logprob_gather_ref(logits: torch.Tensor, tokens, suppress_mask=None) -> torch.Tensor:
    """Chosen-token log-probabilities: transcribe.py:735 (efficient:
    ``logprob[tok]`` per step) / :1292 (naive: ``logprobs[:, step, tok]``).
    ``suppress_mask`` (bool, same shape as logits) plays the role of the
    in-place -inf masking done by the logit filters at :872-874."""
    x = torch.as_tensor(logits).float().clone()
    if suppress_mask is not None:
        x[torch.as_tensor(suppress_mask)] = -np.inf
    lp = logprobs_ref(x)
    idx = torch.as_tensor(tokens, dtype=torch.long).reshape(-1, 1)
    return lp.gather(-1, idx).squeeze(-1)


def confidence_ref(logprobs) -> float:
    """round(exp(mean(logprobs)), 3); empty -> 0.0  (transcribe.py:984-989,
    993-995; naive :1295-1300).  ``logprobs``: 1-D tensor/list of fp32."""
    lp = torch.as_tensor(logprobs, dtype=torch.float32)
    if lp.numel() == 0:
        return 0.0
    return round(lp.mean().exp().item(), 3)


def confidence_raw_ref(logprobs) -> float:
    """Same as confidence_ref before the reference's round(,3) (parity is
    compared before rounding: BASELINE.md section 2)."""
    lp = torch.as_tensor(logprobs, dtype=torch.float32)
    if lp.numel() == 0:
        return 0.0
    return lp.mean().exp().item()


# --------------------------------------------------------------------------
# Log-mel front end (openai-whisper audio.log_mel_spectrogram, called at
# transcribe.py:1213-1214; constants mirrored at transcribe.py:44-47).
# openai-whisper is absent: its published algorithm is restated with the same
# torch calls (torch.stft centre/reflect, periodic hann, |.|^2, slaney mel,
# log10(clamp 1e-10), max(x, x.max()-8), (x+4)/4).
# --------------------------------------------------------------------------
def mel_filters_ref(n_mels: int = 80) -> torch.Tensor:
    """librosa.filters.mel(sr=16000, n_fft=400, n_mels) restated (slaney scale,
    slaney area norm); cross-checked against transformers.audio_utils in tests."""
    sr, n_fft = SAMPLE_RATE, N_FFT
    fftfreqs = np.linspace(0, sr / 2, n_fft // 2 + 1)

    def hz_to_mel(f):
        f = np.asanyarray(f, dtype=np.float64)
        f_sp = 200.0 / 3
        mels = f / f_sp
        min_log_hz = 1000.0
        min_log_mel = min_log_hz / f_sp
        logstep = np.log(6.4) / 27.0
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)

    def mel_to_hz(m):
        m = np.asanyarray(m, dtype=np.float64)
        f_sp = 200.0 / 3
        min_log_hz = 1000.0
        min_log_mel = min_log_hz / f_sp
        logstep = np.log(6.4) / 27.0
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    mel_f = mel_to_hz(np.linspace(hz_to_mel(0.0), hz_to_mel(sr / 2), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    weights = np.zeros((n_mels, n_fft // 2 + 1))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, None]
    return torch.from_numpy(weights.astype(np.float32))


def log_mel_spectrogram_ref(audio: torch.Tensor, n_mels: int = 80, padding: int = 0) -> torch.Tensor:
    audio = torch.as_tensor(audio, dtype=torch.float32)
    if padding > 0:
        audio = torch.nn.functional.pad(audio, (0, padding))
    window = torch.hann_window(N_FFT)
    stft = torch.stft(audio, N_FFT, HOP_LENGTH, window=window, return_complex=True)
    magnitudes = stft[..., :-1].abs() ** 2
    mel_spec = mel_filters_ref(n_mels) @ magnitudes
    log_spec = torch.clamp(mel_spec, min=1e-10).log10()
    log_spec = torch.maximum(log_spec, log_spec.max() - 8.0)
    log_spec = (log_spec + 4.0) / 4.0
    return log_spec


def pad_or_trim_ref(array: torch.Tensor, length: int = N_FRAMES, axis: int = -1) -> torch.Tensor:
    if array.shape[axis] > length:
        array = array.index_select(dim=axis, index=torch.arange(length))
    if array.shape[axis] < length:
        pad = [(0, 0)] * array.ndim
        pad[axis] = (0, length - array.shape[axis])
        array = torch.nn.functional.pad(array, [p for sizes in pad[::-1] for p in sizes])
    return array
