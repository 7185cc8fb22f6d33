"""Deterministic synthetic inputs shared by the tests, the golden-vector
generator and (CPU side) bench.py's cpu_baseline leg.

* ``StubTokenizer``: the real tokenizer (tiktoken vocabulary) is not available
  offline.  The reference's only known-answer test
  (/root/reference/tests/test_transcribe.py:722-902) lists, for every token id it
  uses, the decoded string -- that table is reproduced here as ``KAT_VOCAB`` so
  the KAT can be replayed; every other id gets a deterministic synthetic piece.
* ``synth_qk``: SURVEY.md section 8(d) "kernel-level set K": N(0,1) logits plus
  a +6.0 ridge, 3 frames wide, on a random monotone token->frame staircase.
"""
from __future__ import annotations

import numpy as np

# id -> bytes for the ids appearing in the reference KAT (multilingual vocab)
KAT_VOCAB = {
    220: " ", 6455: "So", 11: ",", 2232: " uh", 286: " I", 2041: " guess", 8660: " wherever", 291: " you",
    808: " come", 493: " up", 365: " with", 445: " just", 718: " let", 505: " us", 458: " know", 13: ".",
    314: " T", 6: "'", 11771: "fa", 17134: "çon", 4666: " nous", 1022: " sur", 875: "la", 2557: " touch",
    68: "e", 6992: " parce", 631: " que", 269: " c", 377: "est", 409: "un", 7282: " sport", 1956: " qui",
    871: " est", 566: " im", 2707: "port", 394: "ant", 256: " t", 622: "ue", 8208: " deux", 517: " un",
    5977: " peut", 7418: " être", 371: " v", 1004: "io", 306: "le", 580: "nt", 12: "-", 9498: "être",
    9505: " q", 84: "u", 8848: "ذ", 7649: "ان", 8717: " ن", 38251: "سم", 11703: "ّ", 3224: "ه",
    805: " 3", 17: "2", 568: " 2", 18: "3", 21: "6", 502: " 1",
}
# " الآ" split inside the last code point (bytes 20 d8a7 d984 d8|a2), as byte-level BPE does
_ARABIC_HEAD = " الآ".encode("utf-8")
KAT_VOCAB_BYTES = {k: v.encode("utf-8") for k, v in KAT_VOCAB.items()}
KAT_VOCAB_BYTES[6024] = _ARABIC_HEAD[:-1]
KAT_VOCAB_BYTES[95] = _ARABIC_HEAD[-1:]

KAT_VOCAB_EN = {764: b" ."}  # whisper >= 20230314 (test_transcribe.py:895)

_SYLL = ["ka", "to", "mi", "ra", "ne", "so", "lu", "vi", "pa", "de", "go", "zu", "fe", "hi", "wo", "ya"]
_PUNCT = [",", ".", "?", "!", ";", ":"]


def _synthetic_piece(tok: int) -> bytes:
    """Deterministic fake vocabulary for ids outside the KAT table:
    ~60% word starts (leading space), ~30% continuations, ~10% punctuation."""
    h = (tok * 2654435761) & 0xFFFFFFFF
    kind = h % 10
    body = _SYLL[(h >> 8) % 16] + _SYLL[(h >> 12) % 16][: 1 + ((h >> 16) % 2)]
    if kind < 6:
        return (" " + body).encode()
    if kind < 9:
        return body.encode()
    return _PUNCT[(h >> 20) % len(_PUNCT)].encode()


class StubTokenizer:
    """Quacks like whisper.tokenizer.Tokenizer for what the alignment path
    touches: eot, sot, timestamp_begin, decode, decode_with_timestamps."""

    def __init__(self, multilingual: bool = True):
        self.multilingual = multilingual
        if multilingual:
            self.eot, self.sot = 50257, 50258
            self.transcribe, self.translate = 50359, 50358
            self.sot_lm, self.sot_prev, self.no_speech, self.no_timestamps = 50360, 50361, 50362, 50363
            self.timestamp_begin = 50364
            self.vocab = dict(KAT_VOCAB_BYTES)
        else:
            self.eot, self.sot = 50256, 50257
            self.transcribe, self.translate = 50358, 50357
            self.sot_lm, self.sot_prev, self.no_speech, self.no_timestamps = 50359, 50360, 50361, 50362
            self.timestamp_begin = 50363
            self.vocab = dict(KAT_VOCAB_BYTES)
            self.vocab.update(KAT_VOCAB_EN)
        self.sot_sequence = (self.sot, self.sot + 1, self.transcribe) if multilingual else (self.sot,)
        self.language = "en"

    def _piece(self, t: int) -> bytes:
        t = int(t)
        if t >= self.timestamp_begin:
            return f"<|{(t - self.timestamp_begin) * 0.02:.2f}|>".encode()
        if t >= self.eot:
            return b""  # special tokens render as "" (test_transcribe.py:881)
        v = self.vocab.get(t)
        return v if v is not None else _synthetic_piece(t)

    def decode_with_timestamps(self, tokens) -> str:
        return b"".join(self._piece(t) for t in tokens).decode("utf-8", errors="replace")

    def decode(self, tokens) -> str:
        return b"".join(self._piece(t) for t in tokens if int(t) < self.timestamp_begin).decode("utf-8", errors="replace")

    def encode(self, text):  # only used by logit-filter style code on " "
        for k, v in self.vocab.items():
            if v == text.encode():
                return [k]
        raise KeyError(text)


# --------------------------------------------------------------------------
def staircase(rng: np.random.RandomState, T: int, F: int) -> np.ndarray:
    """Random non-decreasing token->frame map with first=0-ish, last<F."""
    if T == 1:
        return np.array([rng.randint(0, F)])
    cuts = np.sort(rng.randint(0, F, size=T))
    return cuts


def synth_qk(seed: int, n_heads: int, T: int, n_ctx: int = 1500, ridge: float = 6.0, width: int = 3,
             lo: int = 0, hi: int | None = None, dtype=np.float32) -> np.ndarray:
    """(n_heads, T, n_ctx) fp32 QK logits: N(0,1) + ridge on a staircase inside
    frames [lo,hi)."""
    rng = np.random.RandomState(seed)
    hi = n_ctx if hi is None else hi
    x = rng.standard_normal((n_heads, T, n_ctx)).astype(np.float32)
    centre = lo + staircase(rng, T, max(hi - lo, 1))
    for t in range(T):
        a = max(int(centre[t]) - width // 2, 0)
        b = min(int(centre[t]) + width // 2 + 1, n_ctx)
        x[:, t, a:b] += ridge
    return x.astype(dtype)


def synth_segment_tokens(seed: int, n_text: int, start_frame: int, end_frame: int, tokenizer: StubTokenizer,
                         with_end: bool = True):
    """[<|start|>, text tokens..., <|end|>] with text ids drawn below eot."""
    rng = np.random.RandomState(seed + 7919)
    text = rng.randint(300, 40000, size=n_text).tolist()
    toks = [tokenizer.timestamp_begin + start_frame] + text
    if with_end:
        toks.append(tokenizer.timestamp_begin + end_frame)
    return toks


# T / F distribution measured on the reference goldens (SURVEY.md section 8:
# T p50 11 / p90 30 / p99 223 / max 225; F p50 144 / p90 352 / max 1500)
def draw_real_shapes(seed: int, n: int):
    rng = np.random.RandomState(seed)
    T = np.clip(np.round(np.exp(rng.normal(np.log(11), 0.8, size=n))), 3, 225).astype(int)
    F = np.clip(np.round(np.exp(rng.normal(np.log(144), 0.7, size=n))), 8, 1500).astype(int)
    F = np.maximum(F, T + 1)
    return T, F
