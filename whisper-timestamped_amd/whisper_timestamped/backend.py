"""Access to the speech-recognition backend (openai-whisper).

whisper-timestamped does not contain a Whisper model: it drives
``openai-whisper`` through forward hooks and a few module-level names
(/root/reference/whisper_timestamped/transcribe.py:14,42-58).  The same holds
here; the import is lazy so that the alignment kernels and
``perform_word_alignment`` are usable without the package.
"""
from __future__ import annotations

from contextlib import contextmanager


def whisper():
    try:
        import whisper as _w
    except ImportError as err:  # pragma: no cover - depends on the environment
        raise ImportError("transcribe()/load_model() need the openai-whisper package (or a compatible module "
                          "registered as `whisper`); the alignment kernels themselves do not") from err
    return _w


def whisper_version() -> str:
    return getattr(whisper(), "__version__", "")


def ge_20230306() -> bool:   # segment tokens include the timestamp tokens, model.alignment_heads exists
    return whisper_version() >= "20230306"


@contextmanager
def attention_weights_exposed(enabled=True):
    """openai-whisper >= 20240930 returns qk=None from the fused SDPA path; the
    reference wraps decoding in whisper.model.disable_sdpa() (transcribe.py:49-58,900).
    ``enabled=False``: leave the backend on its fused path (the QK rows are then computed by wt_qk_rows)."""
    w = whisper()
    ctx = getattr(getattr(w, "model", None), "disable_sdpa", None)
    if not enabled or ctx is None or whisper_version() < "20240930":
        yield
    else:
        with ctx():
            yield


def get_tokenizer(model, task="transcribe", language="en"):
    """transcribe.py:1406-1426 (openai-whisper branch).  With ``model.tokenizer_vocab`` (a path) or $WT_TOKENIZER_VOCAB
    set, the vocabulary comes from that ``.tiktoken`` file through this package's own loader (vocab.py: no
    ``tiktoken`` extension needed) instead of the backend's."""
    import os
    vocab = getattr(model, "tokenizer_vocab", None) or os.environ.get("WT_TOKENIZER_VOCAB")
    if vocab:
        from . import vocab as own
        if os.path.isdir(vocab):
            vocab = os.path.join(vocab, "multilingual.tiktoken" if model.is_multilingual else "gpt2.tiktoken")
        return own.get_tokenizer(model.is_multilingual, num_languages=getattr(model, "num_languages", 99), task=task,
                                 language=language, vocab_path=vocab)
    tk = whisper().tokenizer
    try:
        return tk.get_tokenizer(model.is_multilingual, num_languages=getattr(model, "num_languages", 99), task=task,
                                language=language)
    except TypeError:  # older openai-whisper: no num_languages
        return tk.get_tokenizer(model.is_multilingual, task=task, language=language)


_NOT_DECODING_OPTIONS = ("no_speech_threshold", "logprob_threshold", "compression_ratio_threshold",
                         "condition_on_previous_text", "verbose")


def get_logit_filters(model, whisper_options, prompt=None):
    """The logit filters whisper itself applies while sampling, rebuilt for a given prompt so that
    ``sample_begin`` matches (transcribe.py:1371-1404)."""
    w = whisper()
    opts = {k: v for k, v in whisper_options.items() if k not in _NOT_DECODING_OPTIONS}
    if "initial_prompt" in opts:
        first_prompt = opts.pop("initial_prompt")
        if prompt is None:
            prompt = first_prompt
    if prompt is not None:
        opts["prompt"] = prompt
    options = w.DecodingOptions(without_timestamps=False, max_initial_timestamp=1.0, prefix=None, suppress_blank=True, **opts)
    return w.decoding.DecodingTask(model, options).logit_filters


def norm_language(language):
    if language is None:
        return "en"
    return whisper().tokenizer.TO_LANGUAGE_CODE.get(language.lower(), language)


def should_use_space(language) -> bool:
    return norm_language(language) not in ["zh", "ja", "th", "lo", "my", "yue"]


_FRONT_END_OK = {}


def _front_end_matches(original, wt_audio, device, n_mels=80):
    """Once per process and backend function: the HIP front end must reproduce THIS backend's log_mel_spectrogram on a
    probe signal (2 s of noise + a tone; bar 1e-3, observed 3e-5).  A backend with another window / hop / filterbank
    than the one wt_logmel_batch restates is detected here instead of silently shifting every log-mel."""
    key = (getattr(original, "__module__", None), getattr(original, "__qualname__", None), n_mels) \
        if callable(original) and getattr(original, "__qualname__", None) else None    # a stable identity, never id()
    if key is None or key not in _FRONT_END_OK:
        import logging
        import torch
        try:
            g = torch.Generator().manual_seed(0)
            t = torch.arange(32000) / 16000.0
            probe = (0.05 * torch.randn(32000, generator=g) + 0.1 * torch.sin(2 * 3.141592653589793 * 440.0 * t)).float()
            want = original(probe, n_mels).float().cpu()
            got = wt_audio.log_mel_spectrogram(probe, n_mels=n_mels, device=device).float().cpu()
            ok = want.shape == got.shape and float((want - got).abs().max()) <= 1e-3
        except Exception as err:  # noqa: BLE001 -- whatever an unknown backend raises
            ok = False
            logging.getLogger("whisper_timestamped").warning(f"GPU front end self-check could not run ({err})")
        if not ok:
            logging.getLogger("whisper_timestamped").warning(
                "whisper_timestamped: this backend's log_mel_spectrogram differs from the HIP front end; "
                "using the backend's own (efficient.GPU_FRONT_END is ignored)")
        if key is None:
            return ok
        _FRONT_END_OK[key] = ok
    return _FRONT_END_OK[key]


@contextmanager
def gpu_log_mel(device, enabled=True):
    """While decoding, route openai-whisper's own ``log_mel_spectrogram`` call (whisper/transcribe.py: one call
    for the whole file, + 30 s of padding, file-global max clamp) through the HIP front end (wt_logmel_batch):
    the waveform goes to the GPU once and the (n_mels, n_frames) log-mel never exists on the host.
    Same approach as the reference's forward hooks: the backend is instrumented, not modified."""
    import sys
    mod = sys.modules.get("whisper.transcribe")
    if not enabled or mod is None or not hasattr(mod, "log_mel_spectrogram"):
        yield False
        return
    from . import audio as wt_audio
    original = mod.log_mel_spectrogram
    if not _front_end_matches(original, wt_audio, device):
        yield False                      # this backend's front end is not the one wt_logmel_batch restates: keep its own
        return

    def log_mel_on_gpu(audio, n_mels=80, padding=0, device=None):
        if isinstance(audio, str):
            audio = whisper().load_audio(audio)
        return wt_audio.log_mel_spectrogram(audio, n_mels=n_mels, padding=padding, device=_dev)

    _dev = device
    mod.log_mel_spectrogram = log_mel_on_gpu
    try:
        yield True
    finally:
        mod.log_mel_spectrogram = original
