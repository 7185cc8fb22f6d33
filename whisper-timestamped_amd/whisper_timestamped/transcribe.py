"""Public entry point: ``transcribe_timestamped`` (alias ``transcribe``).

Same signature, option checks, strategy selection and result dictionary as the
reference (/root/reference/whisper_timestamped/transcribe.py:79-357); the
word-alignment numerics run on the MI355X (``efficient.py`` / ``naive.py`` ->
``alignment.py`` -> libwtalign.so).
"""
from __future__ import annotations

import logging
import sys

import torch

from . import backend as _backend
from .efficient import transcribe_efficient
from .naive import transcribe_naive
from .naive import get_audio_tensor
from .output import filtered_keys, flatten, remove_keys, write_csv  # noqa: F401  (reference: same module, same names)
from .postprocess import ensure_increasing_positions, remove_last_null_duration_words
from .vad import check_vad_method, remove_non_speech
from .words import AUDIO_TIME_PER_TOKEN, HOP_LENGTH, N_FRAMES, SAMPLE_RATE

logger = logging.getLogger("whisper_timestamped")

# Cross-attention heads that correlate with word timing, per official checkpoint: (n_layers, n_heads, [(layer, head)]).
# Decoded from the packed boolean masks the reference carries at transcribe.py:2343-2357.
ALIGNMENT_HEADS = {
    "tiny.en": (4, 6, [(1, 0), (2, 0), (2, 5), (3, 0), (3, 1), (3, 2), (3, 3), (3, 4)]),
    "tiny": (4, 6, [(2, 2), (3, 0), (3, 2), (3, 3), (3, 4), (3, 5)]),
    "base.en": (6, 8, [(3, 3), (4, 7), (5, 1), (5, 5), (5, 7)]),
    "base": (6, 8, [(3, 1), (4, 2), (4, 3), (4, 7), (5, 1), (5, 2), (5, 4), (5, 6)]),
    "small.en": (12, 12, [(6, 6), (7, 0), (7, 3), (7, 8), (8, 2), (8, 5), (8, 7), (9, 0), (9, 4), (9, 8), (9, 10), (10, 0),
                          (10, 1), (10, 2), (10, 3), (10, 6), (10, 11), (11, 2), (11, 4)]),
    "small": (12, 12, [(5, 3), (5, 9), (8, 0), (8, 4), (8, 7), (8, 8), (9, 0), (9, 7), (9, 9), (10, 5)]),
    "medium.en": (24, 16, [(11, 4), (14, 1), (14, 12), (14, 14), (15, 4), (16, 0), (16, 4), (16, 9), (17, 12), (17, 14),
                           (18, 7), (18, 10), (18, 15), (20, 0), (20, 3), (20, 9), (20, 14), (21, 12)]),
    "medium": (24, 16, [(13, 15), (15, 4), (15, 15), (16, 1), (20, 0), (23, 4)]),
    "large-v1": (32, 20, [(9, 19), (11, 2), (11, 4), (11, 17), (22, 7), (22, 11), (22, 17), (23, 2), (23, 15)]),
    "large-v2": (32, 20, [(10, 12), (13, 17), (16, 11), (16, 12), (16, 13), (17, 15), (17, 16), (18, 4), (18, 11), (18, 19),
                          (19, 11), (21, 2), (21, 3), (22, 3), (22, 9), (22, 12), (23, 5), (23, 7), (23, 13), (25, 5),
                          (26, 1), (26, 12), (27, 15)]),
    "large-v3": (32, 20, [(7, 0), (10, 17), (12, 18), (13, 12), (16, 1), (17, 14), (19, 11), (21, 4), (24, 1), (25, 6)]),
    "large-v3-turbo": (4, 20, [(2, 4), (2, 11), (3, 3), (3, 6), (3, 11), (3, 14)]),
    "turbo": (4, 20, [(2, 4), (2, 11), (3, 3), (3, 6), (3, 11), (3, 14)]),
}

# parameter count (without an untied projection / HF positional table) -> checkpoint family (transcribe.py:2359-2370)
PARAMETERS_TO_MODEL_NAME = {
    37184256: "tiny.en", 37184640: "tiny", 71825408: "base.en", 71825920: "base", 240582144: "small.en",
    240582912: "small", 762320896: "medium.en", 762321920: "medium", 1541384960: "large", 1541570560: "large-v3",
}


def _number_of_parameters(model):
    skip = ("decoder.proj_out.weight", "model.encoder.embed_positions.weight")
    return sum(p.numel() for name, p in model.named_parameters() if name not in skip)


def alignment_heads_for(model_name, num_layers, num_heads):
    L, H, pairs = ALIGNMENT_HEADS[model_name]
    mask = torch.zeros(L, H, dtype=torch.bool)
    for l, h in pairs:
        mask[l, h] = True
    return mask.reshape(num_layers, num_heads).to_sparse()


def get_alignment_heads(model, max_top_layer=3):
    """The model's own ``alignment_heads`` (openai-whisper >= 20230306), else the table keyed by parameter
    count, else None = "all heads of the top layers" (transcribe.py:2372-2387)."""
    if hasattr(model, "alignment_heads"):
        return model.alignment_heads
    n = _number_of_parameters(model)
    if n not in PARAMETERS_TO_MODEL_NAME:
        logger.warning("Could not retrieve alignment heads : taking all attention heads from the top layers")
        return None
    name = PARAMETERS_TO_MODEL_NAME[n]
    if name == "large":
        name = "large-v1" if next(model.parameters())[0, 0, 0] > 0 else "large-v3"
    return alignment_heads_for(name, model.dims.n_text_layer, model.dims.n_text_head)


def is_transformer_model(model):
    return hasattr(model, "generation_config") and hasattr(model, "processor")


def print_timestamped(word):
    fmt = _backend.whisper().utils.format_timestamp
    line = f"[{fmt(word['start'])} --> {fmt(word['end'])}] {word['text']}\n"
    sys.stdout.write(line.encode(sys.getdefaultencoding(), errors="replace").decode())
    sys.stdout.flush()


def load_model(name, device=None, backend="openai-whisper", download_root=None, in_memory=False):
    """transcribe.py:2405-2544: openai-whisper identifiers / ``.pt`` files through whisper.load_model, anything
    else is taken as a HuggingFace-format checkpoint and converted (checkpoint.py)."""
    import os
    if backend == "transformers":
        raise NotImplementedError("backend 'transformers': only openai-whisper models are supported on this path")
    if backend not in ("openai", "openai-whisper"):
        raise ValueError(f"Got unexpected backend {backend}")
    w = _backend.whisper()
    ext = os.path.splitext(name)[-1] if os.path.isfile(name) else None
    if name in w.available_models() or ext == ".pt":
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        return w.load_model(name, device=device, download_root=os.path.join(download_root, "whisper") if download_root else None,
                            in_memory=in_memory)
    from .checkpoint import convert_hf_state_dict, find_checkpoint_files, torch_load
    return convert_hf_state_dict(torch_load(find_checkpoint_files(name, download_root)), device)


def transcribe_timestamped(
        model, audio, language=None, task="transcribe",
        # word-alignment options
        remove_punctuation_from_words=False, compute_word_confidence=True, include_punctuation_in_confidence=False,
        refine_whisper_precision=0.5, min_word_duration=0.02, plot_word_alignment=False,
        word_alignment_most_top_layers=None, remove_empty_words=False, use_backend_timestamps=False,
        # reproducibility, pre-processing, strategy
        seed=1234, vad=False, detect_disfluencies=False, trust_whisper_timestamps=True, naive_approach=False,
        # decoding options handed to the backend
        temperature=0.0, best_of=None, beam_size=None, patience=None, length_penalty=None,
        compression_ratio_threshold=2.4, logprob_threshold=-1.0, no_speech_threshold=0.6, fp16=None,
        condition_on_previous_text=True, initial_prompt=None, suppress_tokens="-1", sample_len=None, verbose=False):
    """Transcribe ``audio`` with ``model`` and add word timestamps / confidences.

    Arguments, defaults and the returned dictionary are those of the reference
    (whisper's result + per-segment ``confidence`` and ``words[{text,start,end,confidence}]``,
    optional ``language_probs``).  ``model`` must live on the GPU.

    Accepted by the signature but NOT supported here (the call raises ``NotImplementedError``):
    ``vad=True / "silero" / "auditok"`` (the detectors are third-party models that need network access -- pass the
    speech islands as ``vad=[(start, end), ...]``, which runs the reference's glue / back-conversion),
    and a HuggingFace ``transformers`` model object as ``model``.  ``plot_word_alignment=True`` shows, a string saves
    (``<prefix>.alignment<NNN>.jpg``, ``<prefix>.VAD.jpg``) the reference's debug figures (plotting.py; needs matplotlib).
    Module switches that trade exact reference arithmetic for speed are listed in INTEGRATION.md ("Switches")."""
    plan = _plan(model, locals())
    model, audio = plan["model"], audio
    if seed is not None:
        torch.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
    vad, naive_approach = plan["vad"], plan["naive_approach"]
    if plot_word_alignment:
        from . import plotting
        plotting.reset()                       # (transcribe.py:300-301: figures are numbered per call)
    if vad is not None:
        audio = get_audio_tensor(audio)
        audio, vad_segments, convert_timestamps = remove_non_speech(audio, method=vad, sample_rate=SAMPLE_RATE,
                                                                    plot=plot_word_alignment, avoid_empty_speech=True)
    else:
        vad_segments = convert_timestamps = None

    if naive_approach:
        transcription, words = transcribe_naive(model, audio, min_word_duration=0.0,
                                                trust_whisper_timestamps=trust_whisper_timestamps,
                                                use_backend_timestamps=use_backend_timestamps,
                                                **plan["alignment_options"], **plan["whisper_options"], **plan["other_options"])
    else:
        transcription, words = transcribe_efficient(model, audio, trust_whisper_timestamps=trust_whisper_timestamps,
                                                    **plan["alignment_options"], **plan["whisper_options"],
                                                    **plan["other_options"])
    return _assemble(transcription, words, plan, vad_segments, convert_timestamps)


def _plan(model, a):
    """Option checks, strategy selection and the option dictionaries of T.py:221-296, from the caller's arguments `a`."""
    refine_whisper_precision, min_word_duration = a["refine_whisper_precision"], a["min_word_duration"]
    word_alignment_most_top_layers = a["word_alignment_most_top_layers"]
    temperature, best_of, beam_size = a["temperature"], a["best_of"], a["beam_size"]
    naive_approach, verbose = a["naive_approach"], a["verbose"]

    steps = refine_whisper_precision / AUDIO_TIME_PER_TOKEN
    assert refine_whisper_precision >= 0 and steps == round(steps), \
        f"refine_whisper_precision must be a positive multiple of {AUDIO_TIME_PER_TOKEN}"
    refine_nframes = round(steps)
    assert min_word_duration >= 0, "min_word_duration must be a positive number"
    assert word_alignment_most_top_layers is None or word_alignment_most_top_layers > 0, \
        "word_alignment_most_top_layers must be a strictly positive number"

    if isinstance(temperature, (list, tuple)) and len(temperature) == 1:
        temperature = temperature[0]
    if isinstance(temperature, (list, tuple)):                 # temperature fallback
        naive_approach = True
    elif temperature > 0 and best_of is not None and best_of > 1:  # random sampling
        naive_approach = True
    if beam_size is not None:                                   # beam search
        naive_approach = True
    if is_transformer_model(model) or a["use_backend_timestamps"]:
        naive_approach = True

    vad = check_vad_method(a["vad"])
    if isinstance(model, str):
        model = load_model(model)
    fp16 = a["fp16"]
    if fp16 is None:
        fp16 = model.device != torch.device("cpu")

    input_stride = N_FRAMES // model.dims.n_audio_ctx
    assert input_stride * HOP_LENGTH / SAMPLE_RATE == AUDIO_TIME_PER_TOKEN

    alignment_heads = get_alignment_heads(model) if word_alignment_most_top_layers is None else None
    if alignment_heads is None and word_alignment_most_top_layers is None:
        word_alignment_most_top_layers = 6

    alignment_options = dict(
        remove_punctuation_from_words=a["remove_punctuation_from_words"], compute_word_confidence=a["compute_word_confidence"],
        include_punctuation_in_confidence=a["include_punctuation_in_confidence"], detect_disfluencies=a["detect_disfluencies"],
        refine_whisper_precision_nframes=refine_nframes, plot_word_alignment=a["plot_word_alignment"],
        word_alignment_most_top_layers=word_alignment_most_top_layers, alignment_heads=alignment_heads)
    whisper_options = dict(
        language=a["language"], task=a["task"], fp16=fp16, temperature=temperature, best_of=best_of, beam_size=beam_size,
        patience=a["patience"], length_penalty=a["length_penalty"], condition_on_previous_text=a["condition_on_previous_text"],
        initial_prompt=a["initial_prompt"], suppress_tokens=a["suppress_tokens"], sample_len=a["sample_len"],
        verbose=verbose if (not vad or verbose is not True) else False,
    )
    other_options = dict(no_speech_threshold=a["no_speech_threshold"], logprob_threshold=a["logprob_threshold"],
                         compression_ratio_threshold=a["compression_ratio_threshold"])
    return dict(model=model, vad=vad, naive_approach=naive_approach, alignment_options=alignment_options,
                whisper_options=whisper_options, other_options=other_options, verbose=verbose,
                refine_whisper_precision=refine_whisper_precision, min_word_duration=min_word_duration,
                remove_empty_words=a["remove_empty_words"], trust_whisper_timestamps=a["trust_whisper_timestamps"])


def _assemble(transcription, words, plan, vad_segments=None, convert_timestamps=None):
    """(transcription, words) of a strategy -> the public result dictionary (T.py:313-357)."""
    verbose, vad, naive_approach = plan["verbose"], plan["vad"], plan["naive_approach"]
    refine_whisper_precision = plan["refine_whisper_precision"]
    if plan["remove_empty_words"]:
        transcription, words = remove_last_null_duration_words(transcription, words, recompute_text=True)

    ensure_increasing_positions(words, min_duration=plan["min_word_duration"] if plan["trust_whisper_timestamps"] else 0)

    segments = transcription["segments"]
    for word in words:
        if verbose and not naive_approach and not vad:
            print_timestamped(word)
        word.pop("tokens", None)
        word.pop("tokens_indices", None)
        word.pop("avg_logprob_reliable", None)
        idx_segment = word.pop("idx_segment")
        assert idx_segment < len(segments), f"Fatal error: Got unexpected segment index {idx_segment} >= {len(segments)}"
        segment = segments[idx_segment]
        if "words" in segment:
            segment["words"].append(word)
        else:
            segment["words"] = [word]
            if refine_whisper_precision:
                segment["start"] = word["start"]
        if refine_whisper_precision:
            segment["end"] = word["end"]

    if vad:                                                     # back to the time axis of the original audio
        for segment in segments:
            for word in segment.get("words", []):
                word["start"], word["end"] = convert_timestamps(word["start"], word["end"])
                if verbose:
                    print_timestamped(word)
            if refine_whisper_precision and len(segment.get("words", [])):
                segment["start"] = segment["words"][0]["start"]
                segment["end"] = segment["words"][-1]["end"]
            else:
                segment["start"], segment["end"] = convert_timestamps(segment["start"], segment["end"])
    if vad_segments is not None:
        transcription["speech_activity"] = [{"start": s, "end": e} for (s, e) in vad_segments]
    return transcription


def transcribe_batch(model, audios, max_streams=32, **options):
    """``[transcribe_timestamped(model, a, **options) for a in audios]`` for INDEPENDENT recordings, with up to
    ``max_streams`` of them stepping through the decoder together (streams.py).  Not in the reference, whose efficient
    strategy decodes one stream (T.py:806): same options, same result dictionaries, one per recording.  Calls the B-stream
    path cannot take (naive strategy: beam search / temperature fallback / best_of; ``vad``) run one recording at a time."""
    import inspect
    sig = inspect.signature(transcribe_timestamped)
    unknown = set(options) - set(sig.parameters)
    assert not unknown, f"transcribe_batch: unknown options {sorted(unknown)}"
    a = {k: p.default for k, p in sig.parameters.items() if p.default is not inspect.Parameter.empty}
    a.update(options)
    audios = list(audios)
    if not audios:
        return []
    plan = _plan(model, a)
    from . import streams
    # (one recording at a time: the model _plan has already loaded -- a name or a path is not read from disk again per recording)
    if not streams.supports(plan["whisper_options"], plan["vad"], plan["naive_approach"], a["plot_word_alignment"]):
        return [transcribe_timestamped(plan["model"], audio, **options) for audio in audios]
    missing = streams.backend_missing()
    if missing:
        logger.warning(f"transcribe_batch: this ASR backend lacks {', '.join(missing)}: decoding one stream at a time")
        return [transcribe_timestamped(plan["model"], audio, **options) for audio in audios]
    if a["seed"] is not None:
        torch.manual_seed(a["seed"])
        torch.cuda.manual_seed_all(a["seed"])
    with streams.paused_gc():          # (thousands of small acyclic objects per stream: reference counting frees them)
        pairs = streams.transcribe_efficient_streams(plan["model"], audios, trust_whisper_timestamps=a["trust_whisper_timestamps"],
                                                     max_streams=max_streams, **plan["alignment_options"],
                                                     **plan["whisper_options"], **plan["other_options"])
        return [_assemble(t, w, plan) for t, w in pairs]


transcribe = transcribe_timestamped
