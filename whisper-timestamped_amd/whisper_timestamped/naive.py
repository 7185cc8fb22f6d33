"""Naive strategy: transcribe first, then re-run the decoder teacher-forced on
every segment (or 30 s window) to get attention weights and log-probabilities.

Behavioural mirror of ``_transcribe_timestamped_naive``
(/root/reference/whisper_timestamped/transcribe.py:1004-1338).  Used for beam
search / temperature fallback / best_of, where decoding is not greedy and the
on-the-fly hooks would see discarded hypotheses.

MI355X data plane: the crop's log-mel comes from the HIP front end
(``wt_logmel_batch``), the selected heads' QK rows stay on the GPU and go
straight to the cost/DTW kernels, and the chosen-token log-probabilities come
from ``wt_logprob_gather_batch`` on the teacher-forced logits (no (T,V)
log-softmax matrix, :1245).
"""
from __future__ import annotations

import logging
import sys

import torch

from . import _lib, audio as wt_audio, backend
from .alignment import AlignmentBatch, head_pairs, prepare_unit
from .confidence import strip_trailing_punctuation
from .words import (AUDIO_SAMPLES_PER_TOKEN, AUDIO_TIME_PER_TOKEN, HOP_LENGTH, N_FRAMES, SAMPLE_RATE, SEGMENT_DURATION,
                    round_confidence)

logger = logging.getLogger("whisper_timestamped")

# trust_whisper_timestamps=False: how many of whisper's (independent) 30 s windows share one launch set in the second
# pass (batched.py).  0 = the reference's shape, one window at a time.
BATCH_WINDOWS = 32


def get_audio_tensor(audio, device="cpu"):
    """transcribe.py:1340-1347"""
    if isinstance(audio, str):
        audio = backend.whisper().load_audio(audio)
    import numpy as np
    if isinstance(audio, np.ndarray):
        audio = torch.Tensor(audio)
    else:
        assert isinstance(audio, torch.Tensor), f"Got unexpected audio of type {type(audio)}"
    return audio.to(device)


def audio_minimum_padding(audio):
    """transcribe.py:1349-1352: the STFT needs more than n_fft/2 samples."""
    if audio.shape[-1] <= 200:
        return wt_audio.pad_or_trim(audio, 201)
    return audio


def _select_heads(captured, pairs, first_row):
    """captured: per hooked layer (1,H,T_all,n_ctx) GPU tensors -> (A_sel, T_all-first_row, n_ctx) GPU tensor."""
    if pairs is None:
        cat = torch.cat(captured)                       # (L,H,T,n_ctx)
        sel = cat.reshape(-1, *cat.shape[-2:])
    else:
        sel = torch.stack([captured[l][0, h] for l, h in pairs])
    return sel[:, first_row:, :].contiguous()


def transcribe_naive(model, audio, *, remove_punctuation_from_words, compute_word_confidence,
                     include_punctuation_in_confidence, refine_whisper_precision_nframes, use_backend_timestamps,
                     alignment_heads, plot_word_alignment, word_alignment_most_top_layers, detect_disfluencies,
                     trust_whisper_timestamps, min_word_duration, **whisper_options):
    w = backend.whisper()
    verbose = whisper_options["verbose"]
    whisper_options["verbose"] = None if verbose is True else verbose
    language = whisper_options["language"]
    refine_sec = refine_whisper_precision_nframes * AUDIO_TIME_PER_TOKEN
    n_blocks = len(model.decoder.blocks)
    top = n_blocks if word_alignment_most_top_layers is None else min(word_alignment_most_top_layers, n_blocks)

    audio = get_audio_tensor(audio)
    audio_duration = audio.shape[-1] / SAMPLE_RATE
    if verbose and language is None and not whisper_options["verbose"]:
        print("Detecting language using up to the first 30 seconds. Use `--language` to specify the language")
    tokenizer = backend.get_tokenizer(model, task=whisper_options["task"], language=language)
    whisper_options["word_timestamps"] = use_backend_timestamps

    language_probs = None

    def hook_language(layer, ins, outs):
        nonlocal language_probs
        if language is None and language_probs is None:
            if outs.shape[1] == 1:
                emb_t = torch.transpose(model.decoder.token_embedding.weight, 0, 1).to(outs[0].dtype)
                lo = tokenizer.sot + 1
                logits = (outs[0][0, :] @ emb_t).float()
                probs = logits[lo:lo + len(tokenizer.all_language_tokens)].softmax(dim=-1)
                language_probs = dict(zip(w.tokenizer.LANGUAGES, probs.tolist()))
            else:
                language_probs = False

    hooks = []
    if model.is_multilingual:
        hooks.append(model.decoder.ln.register_forward_hook(hook_language))
    try:
        model.alignment_heads = alignment_heads
        from .efficient import FUSED_ATTENTION as _fused, GPU_FRONT_END
        with torch.no_grad(), backend.attention_weights_exposed(not _fused), backend.gpu_log_mel(model.device, GPU_FRONT_END):
            transcription = model.transcribe(audio, **whisper_options)
    finally:
        for h in hooks:
            h.remove()
    if verbose and language is None and not whisper_options["verbose"]:
        print(f"Detected language: {w.tokenizer.LANGUAGES[transcription['language']].title()}")
        sys.stdout.flush()

    if transcription.get("segments") and "words" in transcription["segments"][0]:   # the backend made the timestamps
        words = []
        for i_segment, segment in enumerate(transcription["segments"]):
            ws = segment.pop("words", [])
            for word in ws:
                if "word" in word:
                    word["text"] = word.pop("word")
                if "probability" in word:
                    word["confidence"] = round_confidence(word.pop("probability"))
                word["idx_segment"] = i_segment
            words.extend(ws)
        if language_probs:
            transcription["language_probs"] = language_probs
        return transcription, words

    language = backend.norm_language(transcription.get("language", language))
    use_space = backend.should_use_space(language)
    n_mels = model.dims.n_mels if hasattr(model.dims, "n_mels") else 80
    dev = model.device
    _lib.require_gpu(dev)
    pairs = head_pairs(alignment_heads)
    captured = [None] * top
    from .efficient import FUSED_ATTENTION
    from .capture import QKCaptureRing
    fused_q, fused_k = [None] * top, [None] * top
    ring = QKCaptureRing(dev, pairs, top, model.dims.n_text_head, n_ctx=model.dims.n_audio_ctx,
                         capacity=model.dims.n_text_ctx) if FUSED_ATTENTION else None

    hooks = []
    try:
        j = 0
        for i, block in enumerate(model.decoder.blocks):
            if i < n_blocks - top:
                continue
            if FUSED_ATTENTION:     # q / K of the projections: the rows are computed by wt_qk_rows after the forward
                hooks.append(block.cross_attn.query.register_forward_hook(lambda m, i, o, index=j: fused_q.__setitem__(index, o)))
                hooks.append(block.cross_attn.key.register_forward_hook(lambda m, i, o, index=j: fused_k.__setitem__(index, o)))
            else:
                hooks.append(block.cross_attn.register_forward_hook(
                    lambda layer, ins, outs, index=j: captured.__setitem__(index, outs[1])))
            j += 1

        window_tokens, token_to_segment = [], []
        words = []
        previous_end = 0
        segments = transcription["segments"]
        audio_dev = None
        pending = []                       # trust_whisper_timestamps=False: independent windows, aligned as one batch

        def finish_window(ws, word_logprobs, tokens, start, i_segment, segment, token_to_segment, check, last_token_check):
            """What the reference does with a window's words (:1264-1323).  word_logprobs: per word, the CPU fp32
            log-probabilities of its kept tokens (None when confidences are off)."""
            nonlocal previous_end
            segment_logprobs = []
            i_token = 1
            for k, word in enumerate(ws):
                word["start"] = round(word["start"] + start, 2)
                word["end"] = round(word["end"] + start, 2)
                if trust_whisper_timestamps:
                    word.update({"idx_segment": i_segment})
                else:
                    assert i_token < len(tokens)
                    assert not len(word["tokens_indices"]) or word["tokens_indices"][0] == tokens[i_token]
                    word.update({"idx_segment": token_to_segment[i_token]})
                    i_token += len(word["tokens"])
                    while i_token < len(tokens) and tokens[i_token] >= tokenizer.timestamp_begin:
                        i_token += 1
                check.extend(word["tokens_indices"])
                if compute_word_confidence:
                    wl = word_logprobs[k]
                    if len(wl):
                        segment_logprobs.append(wl)
                        conf = wl.mean().exp().item()
                    else:
                        conf = 0
                    word.update({"confidence": round_confidence(conf)})
                words.append(word)
                if verbose:
                    from .transcribe import print_timestamped
                    print_timestamped(word)
            if last_token_check is not None:
                check.append(last_token_check)
            if trust_whisper_timestamps:
                if check != segment["tokens"]:
                    assert len(check) < len(segment["tokens"]), \
                        f"First should be longer by one token: '{tokenizer.decode_with_timestamps(check)}' should include '{tokenizer.decode_with_timestamps(segment['tokens'])}'"
                    assert check[:-1] == segment["tokens"][:len(check) - 1], \
                        f"Got inconsistent tokens: {tokenizer.decode_with_timestamps(check)} != {tokenizer.decode_with_timestamps(segment['tokens'])}"
                    segment["tokens"] = check
                    segment["text"] = tokenizer.decode(segment["tokens"])
            if len(segment_logprobs):
                segment.update({"confidence": round_confidence(torch.cat(segment_logprobs).mean().exp().item())})
            if len(ws):
                previous_end = ws[-1]["end"]

        for i_segment, segment in enumerate(segments):
            start = end = tokens = None
            if trust_whisper_timestamps:
                start, end = segment["start"], segment["end"]
                if end < start:                                     # whisper got the end wrong
                    end = min(audio_duration, start + SEGMENT_DURATION)
                lo_start, hi_start = start - refine_sec, start + refine_sec
                if start >= audio_duration - min_word_duration or (lo_start <= previous_end <= hi_start):
                    start = previous_end                            # decoding restarts at <|0.00|>: be as exact as possible
                else:
                    start = lo_start
                if start > audio_duration - min_word_duration:
                    logger.warning(f"Skipping segment outside of audio duration {audio_duration} (original: {segment['start']}-{segment['end']}, new: {start}-XXX)")
                    continue
                lo_end, hi_end = end - refine_sec, end + refine_sec
                if i_segment < len(segments) - 1:                   # leave room before the next segment
                    hi_end2 = segments[i_segment + 1]["start"] + refine_sec - min_word_duration
                    if hi_end2 >= lo_end:
                        hi_end = min(hi_end2, hi_end)
                end = min(audio_duration, hi_end)
                if end < start + min_word_duration:
                    logger.warning(f"Got super short segment (original from whisper: {segment['start']}-{segment['end']}, new: {start, end})")
                    end = min(audio_duration, start + min_word_duration)
                    if end <= start:
                        logger.warning("Skipping this short segment occuring too close to the end of the audio")
                        continue
                tokens = segment["tokens"]
            else:
                seek = segment["seek"]
                new_tokens = segment["tokens"]
                if not len(new_tokens):
                    continue
                ts0 = tokenizer.timestamp_begin
                if new_tokens[0] < ts0:                              # add the timestamps the alignment needs
                    rel = segment["start"] - (seek * HOP_LENGTH / SAMPLE_RATE)
                    new_tokens = [round(rel * SAMPLE_RATE / AUDIO_SAMPLES_PER_TOKEN) + ts0] + new_tokens
                if new_tokens[-1] < ts0:
                    rel = segment["end"] - (seek * HOP_LENGTH / SAMPLE_RATE)
                    new_tokens = new_tokens + [round(rel * SAMPLE_RATE / AUDIO_SAMPLES_PER_TOKEN) + ts0]
                window_tokens.extend(new_tokens)
                token_to_segment.extend([i_segment] * len(new_tokens))
                next_seek = segments[i_segment + 1]["seek"] if i_segment < len(segments) - 1 else None
                if seek != next_seek:
                    start = float(seek * HOP_LENGTH / SAMPLE_RATE)
                    assert start < audio_duration, f"Got start {start} which is outside of audio duration {audio_duration}"
                    end = min(start + SEGMENT_DURATION, audio_duration)
                    tokens = window_tokens
            if tokens is None or not len(tokens):
                continue

            start_sample = min(round(start * SAMPLE_RATE), audio.shape[-1])
            end_sample = min(round(end * SAMPLE_RATE), audio.shape[-1])

            if not trust_whisper_timestamps and BATCH_WINDOWS and not plot_word_alignment:    # (figures: one window at a time)
                # whisper's 30 s seek groups do not depend on each other (:1197-1202): queue the window, all of them go
                # through mel / encoder / decoder / alignment / confidence gather together (batched.py)
                if audio_dev is None:
                    audio_dev = audio.to(dev).float()                # the recording crosses PCIe once
                from .batched import WindowJob
                pending.append(WindowJob(audio_minimum_padding(audio_dev[start_sample:end_sample]), tokens,
                                         end_sample - start_sample,
                                         tag=(start, i_segment, segment, token_to_segment)))
                window_tokens, token_to_segment = [], []
                continue

            sub_audio = audio_minimum_padding(audio[start_sample:end_sample])
            # log-mel of the crop on the GPU, zero padded / cut to 3000 frames (log_mel_spectrogram + pad_or_trim, :1213-1215)
            n_valid = sub_audio.shape[-1]
            pcm = sub_audio.to(dev).float()
            if n_valid > N_FRAMES * HOP_LENGTH:                      # longer than 30 s: normalise over all frames, then cut
                mfcc = wt_audio.pad_or_trim(wt_audio.log_mel_spectrogram(pcm, n_mels, device=dev), N_FRAMES).unsqueeze(0)
            else:
                nv = torch.tensor([n_valid], dtype=torch.int32, device=dev)
                mfcc = wt_audio.log_mel_batch(wt_audio.pad_or_trim(pcm, N_FRAMES * HOP_LENGTH).reshape(1, -1), nv,
                                              n_mels=n_mels, n_frames=N_FRAMES)

            check = []
            if tokens[0] >= tokenizer.timestamp_begin:
                check.append(tokens[0])
            while tokens[0] >= tokenizer.timestamp_begin:
                tokens = tokens[1:]
                assert len(tokens), "Got transcription with only timestamps!"
            last_token_check = None
            while tokens[-1] >= tokenizer.timestamp_begin:
                last_token_check = tokens[-1]
                tokens = tokens[:-1]

            sot_sequence = tokenizer.sot_sequence
            if language and len(sot_sequence) == 3:
                sot_sequence = (sot_sequence[0], tokenizer.to_language_token(language), sot_sequence[2])
            tokens = [*sot_sequence, tokenizer.timestamp_begin] + tokens
            i_start = len(sot_sequence)

            with torch.no_grad(), backend.attention_weights_exposed(not FUSED_ATTENTION):
                logits = model(mfcc, torch.tensor(tokens, dtype=torch.int32, device=dev).unsqueeze(0))
            logits = logits[0]                                       # (T_all, V), teacher forced; NO logit filters (:1245)

            end_token = tokenizer.timestamp_begin + round(min(N_FRAMES * HOP_LENGTH, end_sample - start_sample) // AUDIO_SAMPLES_PER_TOKEN)
            tokens = tokens[i_start:] + [end_token]
            if FUSED_ATTENTION:
                n_q = fused_q[0].shape[1]
                for index in range(top):
                    ring.write_from_projections(index, fused_q[index], fused_k[index], 0, n_rows=n_q)
                qk_sel = ring.buf[:, i_start - 1:n_q]
            else:
                qk_sel = _select_heads(captured, pairs, i_start - 1)

            unit = prepare_unit(tokens, None, tokenizer, use_space=use_space, mfcc=mfcc,
                                refine_whisper_precision_nframes=refine_whisper_precision_nframes,
                                remove_punctuation_from_words=remove_punctuation_from_words,
                                detect_disfluencies=detect_disfluencies, qk_selected=qk_sel)
            if unit is None:
                ws = []
            else:
                batch = AlignmentBatch(plot=plot_word_alignment)
                if plot_word_alignment:
                    unit.mel = mfcc
                batch.add(unit)
                ws = batch.run()[0]

            # chosen-token log-probabilities of the whole crop: one kernel, one small copy
            steps, toks = [], []
            plan = []
            i_tok = i_start
            for word in ws:
                pieces, ids = word["tokens"], word["tokens_indices"]
                i_end = i_tok + len(pieces)
                if compute_word_confidence:
                    if include_punctuation_in_confidence:            # (sic) the reference strips when this is True (:1288-1291)
                        ids = ids[:len(strip_trailing_punctuation(pieces))]
                    plan.append((len(steps), len(ids)))
                    steps.extend(range(i_tok, i_tok + len(ids)))
                    toks.extend(ids)
                i_tok = i_end
            word_logprobs = None
            if compute_word_confidence:
                lp = torch.empty(0)
                if steps:
                    rows = logits[torch.tensor(steps, device=dev)] if steps != list(range(steps[0], steps[0] + len(steps))) \
                        else logits[steps[0]:steps[0] + len(steps)]
                    lp = _lib.logprob_gather(rows.float().contiguous(), torch.tensor(toks, dtype=torch.int32)).cpu()
                word_logprobs = [lp[off:off + n] for off, n in plan]

            finish_window(ws, word_logprobs, tokens, start, i_segment, segment, token_to_segment, check, last_token_check)
            if not trust_whisper_timestamps:
                window_tokens, token_to_segment = [], []

        if pending:
            from .batched import BatchedAligner, align_windows
            for h in hooks:                # the aligner installs its own (batched) capture hooks
                h.remove()
            hooks = []
            aligner = BatchedAligner(model, tokenizer, language=language, use_space=use_space,
                                     alignment_heads=alignment_heads,
                                     word_alignment_most_top_layers=word_alignment_most_top_layers,
                                     refine_whisper_precision_nframes=refine_whisper_precision_nframes,
                                     remove_punctuation_from_words=remove_punctuation_from_words,
                                     compute_word_confidence=compute_word_confidence,
                                     include_punctuation_in_confidence=include_punctuation_in_confidence,
                                     detect_disfluencies=detect_disfluencies, fused_attention=FUSED_ATTENTION)
            try:
                for res in align_windows(aligner, pending, BATCH_WINDOWS):
                    start, i_segment, segment, t2s = res.tag
                    check = [] if res.first_token_check is None else [res.first_token_check]
                    finish_window(res.words, res.word_logprobs, res.tokens, start, i_segment, segment, t2s, check,
                                  res.last_token_check)
            finally:
                aligner.close()            # its own HIP streams (pipeline.StageSet) and their scratch arenas
    finally:
        for h in hooks:
            h.remove()
    if language_probs:
        transcription["language_probs"] = language_probs
    return transcription, words
