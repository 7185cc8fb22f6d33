"""whisper_double -- TEST INFRASTRUCTURE: a from-scratch stand-in for the
third-party ``openai-whisper`` package (absent from this image, no network).

whisper-timestamped drives openai-whisper through forward hooks
(/root/reference/whisper_timestamped/transcribe.py:883-905) and a handful of
module-level names (``whisper.tokenizer.get_tokenizer``, ``whisper.DecodingOptions``,
``whisper.decoding.DecodingTask``, ``whisper.log_mel_spectrogram`` ...).  This
package implements that surface with the architecture and decode loop described
in SURVEY.md Appendix C (openai-whisper >= 20240930 semantics: cross-attention
returns ``(out, qk)``, segment tokens include timestamp tokens), in plain
PyTorch so the SAME object runs on the CPU (where the reference's own code is
executed against it to produce tests/golden/transcribe_*.json) and on the GPU
(where this repository's ``transcribe`` is executed against it).

There are no trained weights offline: models are random-initialised
(``model.build_model``) and decoding can be SCRIPTED (``decoding.Script``): the
token the sampler returns is taken from a script instead of the arg-max, while
logits / attention / log-probabilities are still the model's own, so every
branch of the hook state machine can be exercised deterministically on both
devices.

``install()`` registers the package as ``sys.modules["whisper"]``.
"""
import sys

__version__ = "20240930"

from . import audio, decoding, model, tokenizer, transcribe as _transcribe_mod, utils  # noqa: E402,F401
from .audio import load_audio, log_mel_spectrogram, pad_or_trim  # noqa: E402,F401
from .decoding import DecodingOptions, DecodingResult, decode, detect_language  # noqa: E402,F401
from .model import ModelDimensions, Whisper, build_model  # noqa: E402,F401
from .transcribe import transcribe  # noqa: E402,F401

_MODELS = ["tiny.en", "tiny", "base.en", "base", "small.en", "small", "medium.en", "medium", "large-v1", "large-v2",
           "large-v3", "large", "large-v3-turbo", "turbo"]


def available_models():
    return list(_MODELS)


def load_model(name, device=None, download_root=None, in_memory=False):
    """openai-whisper's ``load_model`` for a checkpoint FILE (``{"dims": {...}, "model_state_dict": {...}}``, SURVEY.md
    Appendix C); there are no trained checkpoints to fetch by name in this image."""
    import os
    import torch
    if not os.path.isfile(name):
        raise RuntimeError(f"whisper_double has no trained checkpoints (offline image): {name!r} is not a file; use "
                           "whisper_double.build_model(...)")
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    checkpoint = torch.load(name, map_location="cpu")
    net = Whisper(ModelDimensions(**checkpoint["dims"]))
    net.load_state_dict(checkpoint["model_state_dict"])
    return net.to(device).eval()


def install():
    """Make ``import whisper`` resolve to this package (and its submodules)."""
    me = sys.modules[__name__]
    sys.modules["whisper"] = me
    for sub in ("audio", "decoding", "model", "tokenizer", "transcribe", "utils"):
        sys.modules[f"whisper.{sub}"] = sys.modules[f"{__name__}.{sub}"]
    return me
