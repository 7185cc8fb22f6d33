"""Word / segment confidence from chosen-token log-probabilities.

The log-probabilities themselves come from the GPU (``wt_logprob_gather_batch``
through ``LogitsRing.gather`` or ``_lib.logprob_gather``); what is left is
scalar glue on a few floats per word, done with the same torch CPU calls as the
reference so that rounding is identical
(/root/reference/whisper_timestamped/transcribe.py:965-995 efficient strategy,
:1285-1300,1319-1320 naive strategy).
"""
from __future__ import annotations

import torch

from .words import _punctuation, round_confidence


def _exp_mean(logprobs: torch.Tensor) -> float:
    return logprobs.mean().exp().item()


def strip_trailing_punctuation(pieces):
    """Drop trailing pieces whose last character is a punctuation mark ("...", "!!" count), keeping at least one."""
    pieces = list(pieces)
    while len(pieces) > 1 and len(pieces[-1]) and pieces[-1][-1] in _punctuation:
        pieces = pieces[:-1]
    return pieces


def segment_confidences(words, logprobs: torch.Tensor, include_punctuation_in_confidence: bool):
    """Efficient strategy.  ``words``: the segment's word dicts (with "tokens"); ``logprobs``: CPU fp32 tensor of the
    segment's text tokens.  Sets word["confidence"]; returns (segment_confidence, consumed_tokens)."""
    seg_conf = None
    if include_punctuation_in_confidence:
        seg_conf = round_confidence(_exp_mean(logprobs))
    kept = []
    i_end = 0
    for word in words:
        i_start = i_end
        pieces = word["tokens"]
        i_end += len(pieces)
        assert i_end <= len(logprobs), f"Fatal Error: Got out-of-bound index: {i_end} > {len(logprobs)}"
        if include_punctuation_in_confidence:
            wl = logprobs[i_start:i_end]
        else:
            wl = logprobs[i_start:i_start + len(strip_trailing_punctuation(pieces))]
            kept.append(wl)
        word["confidence"] = round_confidence(_exp_mean(wl) if len(wl) else 0.0)
    if not include_punctuation_in_confidence:
        seg_conf = round_confidence(_exp_mean(torch.cat(kept)))
    return seg_conf, i_end
