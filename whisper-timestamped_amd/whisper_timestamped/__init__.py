"""MI355X-native drop-in for the word-alignment hot path of whisper-timestamped.

Mirrors the public names of /root/reference/whisper_timestamped/__init__.py:7-9
(``transcribe`` = ``transcribe_timestamped``, ``load_model``, ``__version__``).
"""
__version__ = "1.15.9+mi355x.r1"

from .alignment import AlignmentBatch, AlignmentUnit, perform_word_alignment, prepare_unit  # noqa: F401
from .words import (split_tokens_on_spaces, split_tokens_on_unicode, round_confidence, round_timestamp)  # noqa: F401
