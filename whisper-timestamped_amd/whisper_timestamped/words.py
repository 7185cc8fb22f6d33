"""Host-side integer/string logic of the alignment path (stays Python, as in the
reference; pinned by the reference's known-answer test).

Reference: /root/reference/whisper_timestamped/transcribe.py
  * frame window from the timestamp tokens ............ :1466-1494
  * BPE tokens -> words (unicode / spaces) ............. :1815-1868
  * punctuation bookkeeping ............................ :1503-1508, :1813
  * jumps -> word begin/end times ...................... :1710-1754, :1783-1793
  * rounding helpers ................................... :1807-1811
"""
from __future__ import annotations

import math
import string
import weakref
from dataclasses import dataclass, field

import numpy as np

N_FRAMES = 3000            # mel frames per 30 s window
N_AUDIO_CTX = N_FRAMES // 2
HOP_LENGTH = 160
SAMPLE_RATE = 16000
AUDIO_SAMPLES_PER_TOKEN = HOP_LENGTH * 2
AUDIO_TIME_PER_TOKEN = AUDIO_SAMPLES_PER_TOKEN / SAMPLE_RATE   # 0.02 s
SEGMENT_DURATION = N_FRAMES * HOP_LENGTH / SAMPLE_RATE        # 30.0 s
DISFLUENCY_MARK = "[*]"

_punctuation = "".join(c for c in string.punctuation if c not in ["-", "'"]) + "。，！？：”、…"


# Debug switch: hand out confidences BEFORE the reference's round(, 3) (transcribe.py:1807-1808), so that parity can
# be asserted at 1e-4 on the raw value (BASELINE.json north_star) instead of at one rounding step.
RAW_CONFIDENCE = False


def _round_like_numpy(x, scale):
    """round(numpy.float64, n) IS numpy's rounding -- multiply by 10**n, round half to even, divide (transcribe.py:1807-1811
    applies the builtin to whatever type it holds, and the word times are numpy scalars) -- which differs from the builtin's
    correctly rounded result for a Python float.  The same three IEEE operations on Python floats: 0.24 us instead of the
    2.2 us of numpy.float64.__round__ (a twentieth of a 128-stream decoder loop's host time); bit-identical, the sign of a
    zero result included (tests/test_host_cpu.py)."""
    y = float(x) * scale
    if y != y or y in (math.inf, -math.inf):
        return np.float64(y)
    r = float(round(y))
    if r == 0.0:
        r = math.copysign(0.0, y)
    return np.float64(r / scale)


def round_confidence(x):
    if RAW_CONFIDENCE:
        return x
    return _round_like_numpy(x, 1000.0) if type(x) is np.float64 else round(x, 3)


def round_timestamp(x):
    return _round_like_numpy(x, 100.0) if type(x) is np.float64 else round(x, 2)


def frame_window(tokens, timestamp_begin: int, refine_nframes: int = 0, describe=lambda: ""):
    """Frame window [start, end) of one segment, or None when the segment is
    empty.  Error behaviour as the reference: RuntimeError for a missing start
    timestamp or a non-positive duration."""
    n = len(tokens)
    assert n > 1, f"Got unexpected sequence of tokens of length {n} {describe()}"
    first = tokens[0] - timestamp_begin
    last = tokens[-1] - timestamp_begin
    if first < 0:
        raise RuntimeError(f"Missing start token in: {describe()}")
    if last < 0:                       # decoder stuck: no closing timestamp
        last = N_AUDIO_CTX
    if last == first and refine_nframes == 0:
        return None
    last = min(N_AUDIO_CTX, max(last, first + n))   # minimal duration, issue #67 upstream
    if refine_nframes > 0:
        first = max(first - refine_nframes, 0)
        last = min(last + refine_nframes, N_AUDIO_CTX)
    if last <= first:
        raise RuntimeError(f"Got segment with null or negative duration {describe()}: {first} {last}")
    return first, last


@dataclass
class WordGroups:
    words: list = field(default_factory=list)
    pieces: list = field(default_factory=list)     # decoded string of each token, per word
    ids: list = field(default_factory=list)        # token ids per word

    def as_tuple(self):
        return self.words, self.pieces, self.ids


_SINGLE_TOKEN_TEXTS = weakref.WeakKeyDictionary()


def _single_token_texts(tokenizer):
    """Per tokenizer OBJECT: token id -> its decoded text (a tokenizer's vocabulary does not change).  The cache lives
    on the object, or in a weak dictionary keyed by it; a tokenizer that allows neither (no __dict__, no weak
    references) gets no cache -- never one keyed on id(), which a later tokenizer with another vocabulary can inherit."""
    try:
        return tokenizer.__dict__.setdefault("_wt_single_token_texts", {})
    except (AttributeError, TypeError):          # frozen / slotted tokenizer objects
        pass
    try:
        cache = _SINGLE_TOKEN_TEXTS.get(tokenizer)
        if cache is None:
            cache = _SINGLE_TOKEN_TEXTS[tokenizer] = {}
        return cache
    except TypeError:                            # not weak-referenceable / unhashable
        return {}


def split_tokens_on_unicode(tokens, tokenizer, remove_punctuation_from_words=False, isolate_punctuations=False):
    """Group tokens into the smallest units that decode to valid unicode, gluing
    a lone punctuation to the previous unit (unless it follows a timestamp)."""
    g = WordGroups()
    ts0, eot = tokenizer.timestamp_begin, tokenizer.eot
    run = []
    single = _single_token_texts(tokenizer)
    for t in tokens:
        run.append(t)
        if len(run) == 1:              # the common case: one decode per DISTINCT token id per tokenizer, not per occurrence
            shown = single.get(t)
            if shown is None:
                shown = single[t] = tokenizer.decode_with_timestamps([t] if (t < eot or t >= ts0) else [])
        else:
            shown = tokenizer.decode_with_timestamps([x for x in run if x < eot or x >= ts0])
        if "\ufffd" in shown:
            continue                   # incomplete multi-byte character: keep accumulating
        pieces = [""] * (len(run) - 1) + [shown]
        stripped = shown.strip()
        glue = (not isolate_punctuations) and bool(stripped) and stripped in _punctuation
        if glue and g.ids and g.ids[-1][-1] >= ts0:
            glue = False               # never glue onto a timestamp token
        if glue:
            if not g.words:
                g.words.append("")
                g.pieces.append([])
            if not remove_punctuation_from_words:
                g.words[-1] += shown
            g.pieces[-1].extend(pieces)
            g.ids[-1].extend(run)      # (IndexError on a leading punctuation, like the reference)
        else:
            g.words.append(shown)
            g.pieces.append(pieces)
            g.ids.append(run)
        run = []
    return g.as_tuple()


def split_tokens_on_spaces(tokens, tokenizer, remove_punctuation_from_words=False):
    """Merge the unicode units into space-delimited words."""
    units, unit_pieces, unit_ids = split_tokens_on_unicode(
        tokens, tokenizer, remove_punctuation_from_words=remove_punctuation_from_words)
    ts0 = tokenizer.timestamp_begin
    is_ts = [ids[0] >= ts0 for ids in unit_ids]
    blank = [not u.strip() for u in units]
    g = WordGroups()
    last = len(units) - 1
    for i, unit in enumerate(units):
        text = unit.strip()
        after_ts = i > 0 and is_ts[i - 1]
        before_ts = i < last and is_ts[i + 1]
        after_blank = i > 0 and blank[i - 1]
        leading_space = unit.startswith(" ") and not blank[i]
        punct = (not blank[i]) and text in _punctuation
        opens_word = is_ts[i] or (not after_blank and (
            after_ts or (leading_space and not punct) or (blank[i] and not before_ts)))
        if opens_word:
            g.words.append(text)
            g.pieces.append(unit_pieces[i])
            g.ids.append(unit_ids[i])
        else:
            g.words[-1] = g.words[-1] + text
            g.pieces[-1].extend(unit_pieces[i])
            g.ids[-1].extend(unit_ids[i])
    return g.as_tuple()


def trailing_punctuation_counts(word_pieces, include_punctuation_in_timing=False):
    """1 for a multi-token word whose last piece is a punctuation mark."""
    counts = [1 if (len(p) > 1 and p[-1] in _punctuation) else 0 for p in word_pieces]
    if include_punctuation_in_timing:
        counts[:-2] = [0] * (len(counts) - 2)
    return counts


def words_from_jumps(jumps, jumps_start, words, word_pieces, word_ids, punct_counts, start_time,
                     refine_nframes, unfinished_decoding, disfluences=None):
    """Word begin/end times from the DTW jumps (frame where each token starts).

    jumps / jumps_start: int arrays of length T+1.  Returns the reference's list of
    dict(text, start, end, tokens, tokens_indices)."""
    sizes = [len(p) for p in word_pieces]
    bounds = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    begins = np.asarray(jumps_start)[bounds[:-1]] * AUDIO_TIME_PER_TOKEN
    ends = np.asarray(jumps)[bounds[1:] - np.asarray(punct_counts, dtype=np.int64)] * AUDIO_TIME_PER_TOKEN
    words, word_pieces, word_ids = list(words), list(word_pieces), list(word_ids)

    if disfluences:
        inserts = []
        tok0 = 0
        for w, size in enumerate(sizes[:-1]):
            if tok0 in disfluences and w > 0:
                b, e = disfluences[tok0]
                inserts.append((w, b * AUDIO_TIME_PER_TOKEN, e * AUDIO_TIME_PER_TOKEN))
            tok0 += size
        for w, b, e in reversed(inserts):
            words.insert(w, DISFLUENCY_MARK)
            word_pieces.insert(w, [])
            word_ids.insert(w, [])
            begins = np.insert(begins, w, b)
            ends = np.insert(ends, w, e)

    if not refine_nframes:             # the timestamp tokens carry the segment bounds
        begins[1] = begins[0]
        ends[-2] = ends[-1]
    keep = slice(1, None) if unfinished_decoding else slice(1, -1)
    # round(numpy.float64, 2) IS numpy's rounding (x.__round__ -> ndarray.round: multiply, rint, divide), which is what
    # the reference applies value by value (transcribe.py:1788-1789, 1810-1811); done here on the whole array at once
    # (10 us per scalar call otherwise: most of the host time of a window)
    starts = np.round(np.asarray(begins[keep], dtype=np.float64) + start_time, 2)
    stops = np.round(np.asarray(ends[keep], dtype=np.float64) + start_time, 2)
    out = []
    for text, b, e, pieces, ids in zip(words[keep], starts, stops, word_pieces[keep], word_ids[keep]):
        if text.startswith("<|"):
            continue
        out.append(dict(text=text, start=b, end=e, tokens=pieces, tokens_indices=ids))
    return out
