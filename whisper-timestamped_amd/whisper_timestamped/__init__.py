"""MI355X-native drop-in for the word-alignment hot path of whisper-timestamped.

Mirrors the public names of /root/reference/whisper_timestamped/__init__.py:7-9
(``transcribe`` = ``transcribe_timestamped``, ``load_model``, ``__version__``).
"""
__version__ = "1.15.9+mi355x.r1"

from .alignment import AlignmentBatch, AlignmentUnit, perform_word_alignment, prepare_unit  # noqa: F401
from .words import (split_tokens_on_spaces, split_tokens_on_unicode, round_confidence, round_timestamp)  # noqa: F401
from .capture import LogitsRing, QKCaptureRing  # noqa: F401
from .postprocess import ensure_increasing_positions, remove_last_null_duration_words  # noqa: F401
from .transcribe import get_alignment_heads, load_model, transcribe, transcribe_timestamped  # noqa: F401
