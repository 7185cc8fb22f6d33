"""MI355X-native drop-in for the word-alignment hot path of whisper-timestamped.

Mirrors the public names of /root/reference/whisper_timestamped/__init__.py:7-9
(``transcribe`` = ``transcribe_timestamped``, ``load_model``, ``__version__``).
"""
__version__ = "1.15.9+mi355x.r6"

from .alignment import AlignmentBatch, AlignmentUnit, perform_word_alignment, prepare_unit  # noqa: F401
from .words import (split_tokens_on_spaces, split_tokens_on_unicode, round_confidence, round_timestamp)  # noqa: F401
from .capture import LogitsRing, QKCaptureRing  # noqa: F401
from .postprocess import ensure_increasing_positions, remove_last_null_duration_words  # noqa: F401
from .transcribe import get_alignment_heads, load_model, transcribe, transcribe_batch, transcribe_timestamped  # noqa: F401
from .pipeline import ChunkBatch, HotPathPipeline, StageSet  # noqa: F401

# The reference re-exports a set of openai-whisper names (its __init__.py:1-5).  openai-whisper is an optional,
# lazily imported backend here, so they are resolved on first access; `audio`, `log_mel_spectrogram`, `pad_or_trim`
# and `load_audio` resolve to this package's GPU front end (same names, same results, computed on the MI355X).
_FROM_WHISPER = {"available_models", "decoding", "model", "normalizers", "tokenizer", "utils", "DecodingOptions",
                 "DecodingResult", "decode", "detect_language", "Whisper", "ModelDimensions", "_download", "_MODELS"}
from .audio import load_audio, log_mel_spectrogram, pad_or_trim  # noqa: E402,F401


def __getattr__(name):
    if name in _FROM_WHISPER:
        from .backend import whisper as _w
        return getattr(_w(), name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
