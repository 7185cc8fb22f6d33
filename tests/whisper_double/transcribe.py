"""whisper.transcribe stand-in: the 30 s window / seek loop around DecodingTask."""
import warnings

import numpy as np
import torch

from .audio import FRAMES_PER_SECOND, HOP_LENGTH, N_FRAMES, N_SAMPLES, SAMPLE_RATE, log_mel_spectrogram, pad_or_trim
from .decoding import DecodingOptions
from .tokenizer import LANGUAGES, get_tokenizer
from .utils import exact_div, format_timestamp


def transcribe(model, audio, *, verbose=None, temperature=(0.0, 0.2, 0.4, 0.6, 0.8, 1.0), compression_ratio_threshold=2.4,
               logprob_threshold=-1.0, no_speech_threshold=0.6, condition_on_previous_text=True, initial_prompt=None,
               carry_initial_prompt=False, word_timestamps=False, prepend_punctuations="\"'“¿([{-",
               append_punctuations="\"'.。,，!！?？:：”)]}、", clip_timestamps="0", hallucination_silence_threshold=None,
               **decode_options):
    assert not word_timestamps, "whisper_double does not implement whisper's own word_timestamps"
    dtype = torch.float16 if decode_options.get("fp16", True) else torch.float32
    if model.device == torch.device("cpu"):
        if dtype == torch.float16:
            warnings.warn("FP16 is not supported on CPU; using FP32 instead")
            dtype = torch.float32
    if dtype == torch.float32:
        decode_options["fp16"] = False

    mel = log_mel_spectrogram(audio, model.dims.n_mels, padding=N_SAMPLES)     # + 30 s of silence for slicing
    content_frames = mel.shape[-1] - N_FRAMES
    content_duration = float(content_frames * HOP_LENGTH / SAMPLE_RATE)

    if decode_options.get("language", None) is None:
        if not model.is_multilingual:
            decode_options["language"] = "en"
        else:
            if verbose:
                print("Detecting language using up to the first 30 seconds. Use `--language` to specify the language")
            mel_segment = pad_or_trim(mel, N_FRAMES).to(model.device).to(dtype)
            _, probs = model.detect_language(mel_segment)
            decode_options["language"] = max(probs, key=probs.get)
            if verbose is not None:
                print(f"Detected language: {LANGUAGES[decode_options['language']].title()}")

    language = decode_options["language"]
    task = decode_options.get("task", "transcribe")
    tokenizer = get_tokenizer(model.is_multilingual, num_languages=model.num_languages, language=language, task=task)

    if isinstance(clip_timestamps, str):
        clip_timestamps = [float(ts) for ts in (clip_timestamps.split(",") if clip_timestamps else [])]
    seek_points = [round(ts * FRAMES_PER_SECOND) for ts in clip_timestamps]
    if len(seek_points) == 0:
        seek_points.append(0)
    if len(seek_points) % 2 == 1:
        seek_points.append(content_frames)
    seek_clips = list(zip(seek_points[::2], seek_points[1::2]))

    def decode_with_fallback(segment):
        temperatures = [temperature] if isinstance(temperature, (int, float)) else temperature
        decode_result = None
        for t in temperatures:
            kwargs = {**decode_options}
            if t > 0:
                kwargs.pop("beam_size", None)       # sampling: no beam search
                kwargs.pop("patience", None)
            else:
                kwargs.pop("best_of", None)         # greedy / beam: no best_of
            decode_result = model.decode(segment, DecodingOptions(**kwargs, temperature=t))
            needs_fallback = False
            if compression_ratio_threshold is not None and decode_result.compression_ratio > compression_ratio_threshold:
                needs_fallback = True               # too repetitive
            if logprob_threshold is not None and decode_result.avg_logprob < logprob_threshold:
                needs_fallback = True               # average log probability too low
            if (no_speech_threshold is not None and decode_result.no_speech_prob > no_speech_threshold
                    and logprob_threshold is not None and decode_result.avg_logprob < logprob_threshold):
                needs_fallback = False              # silence
            if not needs_fallback:
                break
        return decode_result

    clip_idx = 0
    seek = seek_clips[clip_idx][0]
    input_stride = exact_div(N_FRAMES, model.dims.n_audio_ctx)          # mel frames per output token: 2
    time_precision = input_stride * HOP_LENGTH / SAMPLE_RATE           # 0.02 s
    all_tokens, all_segments = [], []
    prompt_reset_since = 0
    if initial_prompt is not None:
        initial_prompt_tokens = tokenizer.encode(" " + initial_prompt.strip())
        all_tokens.extend(initial_prompt_tokens)
    else:
        initial_prompt_tokens = []

    def new_segment(*, start, end, tokens, result):
        tokens = tokens.tolist()
        text_tokens = [token for token in tokens if token < tokenizer.eot]
        return {"seek": seek, "start": start, "end": end, "text": tokenizer.decode(text_tokens), "tokens": tokens,
                "temperature": result.temperature, "avg_logprob": result.avg_logprob,
                "compression_ratio": result.compression_ratio, "no_speech_prob": result.no_speech_prob}

    while clip_idx < len(seek_clips):
        seek_clip_start, seek_clip_end = seek_clips[clip_idx]
        if seek < seek_clip_start:
            seek = seek_clip_start
        if seek >= seek_clip_end:
            clip_idx += 1
            if clip_idx < len(seek_clips):
                seek = seek_clips[clip_idx][0]
            continue
        time_offset = float(seek * HOP_LENGTH / SAMPLE_RATE)
        segment_size = min(N_FRAMES, content_frames - seek, seek_clip_end - seek)
        mel_segment = mel[:, seek: seek + segment_size]
        segment_duration = segment_size * HOP_LENGTH / SAMPLE_RATE
        mel_segment = pad_or_trim(mel_segment, N_FRAMES).to(model.device).to(dtype)

        decode_options["prompt"] = all_tokens[prompt_reset_since:]
        result = decode_with_fallback(mel_segment)
        tokens = torch.tensor(result.tokens)

        if no_speech_threshold is not None:
            should_skip = result.no_speech_prob > no_speech_threshold
            if logprob_threshold is not None and result.avg_logprob > logprob_threshold:
                should_skip = False                 # confident enough: keep it whatever no_speech_prob says
            if should_skip:
                seek += segment_size
                continue

        current_segments = []
        timestamp_tokens = tokens.ge(tokenizer.timestamp_begin)
        single_timestamp_ending = timestamp_tokens[-2:].tolist() == [False, True]
        consecutive = torch.where(timestamp_tokens[:-1] & timestamp_tokens[1:])[0]
        consecutive.add_(1)
        if len(consecutive) > 0:                    # split the window at every <|t|><|t|> pair
            slices = consecutive.tolist()
            if single_timestamp_ending:
                slices.append(len(tokens))
            last_slice = 0
            for current_slice in slices:
                sliced = tokens[last_slice:current_slice]
                t0 = sliced[0].item() - tokenizer.timestamp_begin
                t1 = sliced[-1].item() - tokenizer.timestamp_begin
                current_segments.append(new_segment(start=time_offset + t0 * time_precision,
                                                    end=time_offset + t1 * time_precision, tokens=sliced, result=result))
                last_slice = current_slice
            if single_timestamp_ending:
                seek += segment_size                # no speech after the last timestamp
            else:
                last_timestamp_pos = tokens[last_slice - 1].item() - tokenizer.timestamp_begin
                seek += last_timestamp_pos * input_stride
        else:
            duration = segment_duration
            timestamps = tokens[timestamp_tokens.nonzero().flatten()]
            if len(timestamps) > 0 and timestamps[-1].item() != tokenizer.timestamp_begin:
                last_timestamp_pos = timestamps[-1].item() - tokenizer.timestamp_begin
                duration = last_timestamp_pos * time_precision
            current_segments.append(new_segment(start=time_offset, end=time_offset + duration, tokens=tokens, result=result))
            seek += segment_size

        if verbose:
            for segment in current_segments:
                line = f"[{format_timestamp(segment['start'])} --> {format_timestamp(segment['end'])}] {segment['text']}"
                print(line)

        for segment in current_segments:            # instantaneous or text-less segments are cleared
            if segment["start"] == segment["end"] or segment["text"].strip() == "":
                segment["text"] = ""
                segment["tokens"] = []
                segment["words"] = []

        all_segments.extend([{"id": i, **segment} for i, segment in enumerate(current_segments, start=len(all_segments))])
        all_tokens.extend([token for segment in current_segments for token in segment["tokens"]])
        if not condition_on_previous_text or result.temperature > 0.5:
            prompt_reset_since = len(all_tokens)    # do not feed the prompt tokens if a high temperature was used

    return dict(text=tokenizer.decode(all_tokens[len(initial_prompt_tokens):]), segments=all_segments, language=language)
