"""transcribe()-level leg: the batched SECOND PASS of the naive strategy (whisper_timestamped/batched.py), teacher forced,
32 synthetic 30 s chunks per launch set, with the same chunks through the reference-shaped CPU path beside it."""
import os
import sys
import time

import numpy as np
import torch

from .common import log, make_emitter


# --------------------------------------------------------------------------------------------------- transcribe() level
E2E_SEGMENTS = [(0, 280), (300, 560), (580, 900), (920, 1200), (1220, 1480)]    # 5 timestamped segments per window
E2E_TEXT_PER_SEGMENT = 17                                                       # ~86 text tokens per window (SURVEY 8d set M)


def e2e_transcript(tokenizer, seed):
    """The fixed synthetic transcript of one 30 s window, as whisper hands a window's tokens to the naive strategy with
    trust_whisper_timestamps=False: <|s|> text <|e|><|s'|> text <|e'|> ..."""
    rs = np.random.RandomState(seed)
    ts0 = tokenizer.timestamp_begin
    banned = set(getattr(tokenizer, "non_speech_tokens", ())) | {220}
    toks = []
    for s, e in E2E_SEGMENTS:
        text = [int(t) for t in rs.randint(300, 40000, size=E2E_TEXT_PER_SEGMENT)]
        toks += [ts0 + s] + [t if t not in banned else 300 for t in text] + [ts0 + e]
    return toks


def e2e_cpu_reference_window(W, model_cpu, tokenizer, heads, pcm, window_tokens):
    """One window the way the reference's naive loop does it (transcribe.py:1204-1300), on the CPU through oracle/:
    torch.stft log-mel, the model unfused with every hooked layer's QK observed, log_softmax of the whole (T, V)
    block, the oracle's perform_word_alignment, a Python loop of logprobs[:, step, tok] reads."""
    from oracle import align_ref as O
    import torch.nn.functional as F
    ts0 = tokenizer.timestamp_begin
    mel = O.pad_or_trim_ref(O.log_mel_spectrogram_ref(pcm, model_cpu.dims.n_mels), 3000).unsqueeze(0)
    toks = list(window_tokens)
    while toks[0] >= ts0:
        toks = toks[1:]
    while toks[-1] >= ts0:
        toks = toks[:-1]
    sot = tokenizer.sot_sequence
    if len(sot) == 3:
        sot = (sot[0], tokenizer.to_language_token("en"), sot[2])
    toks = [*sot, ts0] + toks
    i_start = len(sot)
    att = [None] * len(model_cpu.decoder.blocks)
    hooks = [blk.cross_attn.register_forward_hook(lambda m, i, o, k=k: att.__setitem__(k, o[-1]))
             for k, blk in enumerate(model_cpu.decoder.blocks)]
    try:
        with torch.no_grad(), W.model.disable_sdpa():
            logprobs = F.log_softmax(model_cpu(mel, torch.tensor(toks, dtype=torch.int32).unsqueeze(0)), dim=-1)
    finally:
        for h in hooks:
            h.remove()
    end_token = ts0 + round(min(480000, pcm.shape[-1]) // 320)
    toks = toks[i_start:] + [end_token]
    att = [w[:, :, i_start - 1:, :] for w in att]
    ws = O.perform_word_alignment_ref(toks, att, tokenizer, use_space=True, alignment_heads=np.asarray(heads), mfcc=mel,
                                      refine_whisper_precision_nframes=25, detect_disfluencies=False)
    for word in ws:
        ids = word["tokens_indices"]
        lp = [logprobs[:, step, tok] for step, tok in zip(range(i_start, i_start + len(ids)), ids)]
        i_start += len(word["tokens"])
        word["confidence_raw"] = torch.cat(lp).mean().exp().item() if lp else 0.0
        word["mean_logprob_raw"] = torch.cat(lp).mean().item() if lp else None
    return ws


def run_e2e(dev, args, leg, emit):
    """audio-seconds transcribed-with-word-timestamps per second at the transcribe() level (SURVEY 8d "End-to-end
    audio-s/s"): whisper-base, 32 synthetic 30 s chunks per launch set, teacher-forced transcript.  One leg per child
    process: "fp32" (the CPU reference's arithmetic; also the CPU e2e baseline and the word parity against it) or
    "fp16" (half-precision activations, eager).  `emit` publishes
    what has been measured so far: a fault later in the leg cannot take it back."""
    import whisper_double as W          # tests/whisper_double: stand-in for openai-whisper (absent from this image)
    W.install()
    from whisper_timestamped.alignment import head_pairs
    from whisper_timestamped.batched import BatchedAligner, WindowJob, align_windows
    from whisper_timestamped.transcribe import get_alignment_heads
    n_per, steps = args.e2e_windows, args.e2e_steps
    name = args.e2e_model
    model = W.build_model(name, seed=0, device=dev)
    if hasattr(model, "alignment_heads"):
        del model.alignment_heads                          # -> the published whisper-base heads (parameter-count table)
    heads = head_pairs(get_alignment_heads(model))
    tokenizer = W.tokenizer.get_tokenizer(True, language="en", task="transcribe",
                                          **({"num_languages": 100} if model.dims.n_vocab >= 51866 else {}))
    g = torch.Generator(device=dev).manual_seed(4321)
    pcm = torch.randn((n_per, 480000), generator=g, device=dev) * 0.1
    transcripts = [e2e_transcript(tokenizer, 100 + k) for k in range(n_per)]
    jobs = [WindowJob(pcm[k % n_per], transcripts[k % n_per], 480000, tag=k) for k in range(n_per * steps)]
    out = {}

    def timed(aligner):
        list(align_windows(aligner, jobs[:n_per], n_per))                    # warm-up (allocations, GEMM plans)
        torch.cuda.synchronize()
        aligner.timeline = []
        t0 = time.perf_counter()
        res = list(align_windows(aligner, jobs, n_per))
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        tl = aligner.timeline
        aligner.timeline = None
        stage = {k: float(np.mean([t[k] for t in tl])) for k in tl[0]} if tl else {}
        span_ms = stage.pop("span", 0.0)                     # first launch -> last kernel of a launch set, idle gaps included
        gpu_ms = sum(stage.values())                         # the kernels alone (an event pair around each stage)
        align_ms = sum(v for k, v in stage.items() if k != "model")
        n_words = sum(len(r.words) for r in res)
        assert all(len(r.words) > 0 for r in res) and n_words > 0
        return res, {"audio_s_per_s": round(30.0 * len(jobs) / el, 1), "ms_per_launch_set": round(el / steps * 1e3, 3),
                     "gpu_kernel_ms_per_launch_set": round(gpu_ms, 3),
                     "gpu_stage_ms": {k: round(v, 3) for k, v in stage.items()},
                     "alignment_share_of_gpu_time": round(align_ms / gpu_ms, 4) if gpu_ms else None,
                     "gpu_span_ms_per_launch_set": round(span_ms, 3),
                     "gpu_busy_fraction_of_wall": round(min(1.0, gpu_ms * steps / (el * 1e3)), 4),
                     "words_per_launch_set": n_words // steps}

    opts = dict(language="en", alignment_heads=torch.tensor(heads), refine_whisper_precision_nframes=25)
    if leg == "fp32":
        out = {"workload": f"whisper-{name} (random init, fp32), {n_per} x 30 s synthetic chunks per launch set, "
                           f"{len(transcripts[0])} window tokens in {len(E2E_SEGMENTS)} timestamped segments, teacher forced "
                           f"(naive strategy, trust_whisper_timestamps=False shape)", "chunks_per_launch": n_per,
               "launch_sets": steps, "alignment_heads": len(heads),
               "what_this_leg_is": "the batched SECOND PASS of the naive strategy (naive_approach=True, trust_whisper_timestamps=False: "
                                   "the transcript is given, the decoder is teacher forced) -- not what transcribe(model, audio) does "
                                   "by default; that is the `default_strategy` object below"}
        res32, fp32 = timed(BatchedAligner(model, tokenizer, **opts))
        out.update(fp32)
        out["dtype"] = "f32 model (the CPU reference's arithmetic), f32 alignment, f64 DTW"
        emit(out)
        if not args.no_cpu_baseline:
            # the same chunks through the reference-shaped CPU path, bounded sample
            model_cpu = W.build_model(name, seed=0, device="cpu")
            pcm_cpu = pcm[:8].cpu()
            done, worst_t, worst_c, worst_l, t0 = 0, 0.0, 0.0, 0.0, time.perf_counter()
            while done < 8:
                ws = e2e_cpu_reference_window(W, model_cpu, tokenizer, heads, pcm_cpu[done], transcripts[done])
                got = res32[done]
                assert [x["text"] for x in got.words] == [x["text"] for x in ws], "GPU and CPU words differ"
                for a, lp, b in zip(got.words, got.word_logprobs, ws):
                    worst_t = max(worst_t, abs(a["start"] - b["start"]), abs(a["end"] - b["end"]))
                    conf = lp.mean().exp().item() if len(lp) else 0.0
                    worst_c = max(worst_c, abs(conf - b["confidence_raw"]))
                    # (a random-init model gives p ~ 1/V: the confidences are ~1e-8 and their difference says nothing;
                    #  the mean log-probabilities they are the exp() of are compared as well)
                    assert (len(lp) == 0) == (b["mean_logprob_raw"] is None)
                    if len(lp):
                        worst_l = max(worst_l, abs(lp.mean().item() - b["mean_logprob_raw"]))
                done += 1
                if time.perf_counter() - t0 > args.e2e_cpu_budget:
                    break
            el = time.perf_counter() - t0
            out["cpu_baseline_e2e"] = {"value": round(30.0 * done / el, 2), "unit": "audio-seconds/s",
                                       "cores": int(torch.get_num_threads()), "kind": "port",
                                       "sample": f"{done} of the same chunks, one at a time as the reference does: torch.stft log-mel, "
                                                 f"the same whisper-base on the CPU with unfused attention and per-layer QK capture, "
                                                 f"log_softmax of the (T, V) block, oracle perform_word_alignment, {el:.1f} s wall"}
            out["parity_vs_cpu_reference_path"] = {"chunks": done, "max_abs_dt_word_s": round(worst_t, 4),
                                                   "max_abs_dconfidence_before_rounding": float(f"{worst_c:.3g}"),
                                                   "max_abs_dmean_logprob_per_word": float(f"{worst_l:.3g}"),
                                                   "bars": {"dt_word_s": 0.02, "dconfidence": 1e-4, "dmean_logprob": 2e-4}}
            assert worst_t <= 0.02 + 1e-9 and worst_c <= 1e-4 and worst_l <= 2e-4, out["parity_vs_cpu_reference_path"]
            out["speedup_vs_cpu_e2e"] = round(out["audio_s_per_s"] / out["cpu_baseline_e2e"]["value"], 1)
            emit(out)
        return out
    # the reference's GPU default is fp16=True (transcribe.py:240-241): the same pipeline with half-precision
    # activations -- whisper keeps LayerNorm in fp32 and casts the other weights per call; here they are cast once.
    for m in model.modules():
        if isinstance(m, (torch.nn.Linear, torch.nn.Conv1d, torch.nn.Embedding)):
            m.half()
    res16, fp16 = timed(BatchedAligner(model, tokenizer, mel_dtype=torch.float16, **opts))
    out["fp16_model"] = fp16
    emit(out)
    return out

