"""transcribe()-level leg: the DEFAULT strategy (the reference's efficient strategy), one stream and B decoder streams,
uniform / ragged / long-form work, the reference-shaped CPU path beside it and the per-word parity checks."""
import os
import sys
import time

import numpy as np
import torch

from .common import log, make_emitter
from wordgaps import (NO_GAPS, PARITY_FAILURES, gaps_ok, gaps_ok_between_batch_sizes, gaps_report, merge_gaps, parity_flag,
                      word_gaps, words_of)


def run_efficient_leg(args, emit):
    """The DEFAULT strategy of transcribe() (the reference's efficient strategy: word alignment on the fly while the
    backend decodes, T.py:359-1001).  Model: the whisper double with PEAKED cross-attention (a monotone ridge on the
    alignment heads, as a trained model has: whisper_double.model.sharpen_cross_attention), scripted transcripts whose
    timestamps follow the ridge (many_helper.peaked_window).
      1_stream          what a caller of the reference's API gets per process: transcribe(model, clip), one decoder stream,
                        one token at a time through the backend's own Python loop;
      B_streams         transcribe_batch(model, clips): B independent recordings stepping through the decoder together
                        (whisper_timestamped/streams.py), B = --e2e-streams (32 = BASELINE configs[1]'s batch), and 4 B --
                        UNIFORM work: 30 s clips, one scripted ~110-token transcript in 5 segments for every stream (every
                        stream finishes in the same decoder call: the lock-step best case);
      ragged_B_streams  the same on RAGGED work: clip lengths U[5, 30] s, a different scripted transcript per stream
                        (2-9 segments, 40-160 tokens), with the driver's streams-per-loop histogram;
      long_form_1h_islands  BASELINE configs[3] at N = 1, uniform and ragged (per-window transcripts drawn per island, so
                        the prompts of windows 2, 3 differ in length from stream to stream under condition_on_previous_text);
      cpu_baseline      the reference-shaped CPU path: the same model on the host cores, unfused attention with per-token QK
                        capture, a second projection + logit filters per token, one synchronous alignment per segment
                        through oracle/ (the reference's shape, T.py:783-793,849-881,544-557), one stream;
      parity            north_star's bar, EVERY word (|dt| <= 0.02 s, |d confidence| <= 1e-4, |d mean log-prob| <= 2e-4):
                        every B-stream leg against one-stream runs of the same recordings, AND recordings of the TIMED
                        ragged / long-form batches against the CPU reference path (parity_vs_cpu_reference_path);
      flat_attention    the round-5 comparison on plain random-init weights (nearly flat attention rows: the DTW has near-ties
                        that the backend's batch-size-dependent GEMM rounding can flip) -- reported, gates nothing."""
    import many_helper as H          # tests/: the whisper double as the model, the scripted transcript
    import whisper_double as W
    from whisper_double.decoding import Script, set_row_scripts, set_script
    from golden import make_golden_transcribe as G
    W.install()
    import whisper_timestamped as wt
    from whisper_timestamped import streams, words
    words.RAW_CONFIDENCE = True      # confidences before the reference's round(, 3): parity is asserted on the raw values
    dev = getattr(args, "e2e_device", "cuda:0")     # (a CPU dry run of this leg's host logic: tools/dry_run_efficient_leg.py)
    load = getattr(args, "e2e_load_model", H.load_base)
    model = load(dev, attention="peaked")
    B = args.e2e_streams
    TS0, EOT = 50364, 50257
    g = torch.Generator().manual_seed(7)
    clips = [(0.05 * torch.randn(30 * 16000, generator=g)).float() for _ in range(4)]
    window = G.window_script(TS0, EOT, [(s, [None] * n, e) for s, n, e in H.peaked_segments([n for _, n, _ in H.SEGMENTS])], "eot")
    out = {"workload": "whisper-base dims, fp32, PEAKED cross-attention on the alignment heads (random init otherwise), synthetic "
                       "clips, scripted transcripts, transcribe() with its defaults (efficient strategy, greedy, "
                       "condition_on_previous_text=True); uniform legs: 30 s clips, one ~110-token transcript in 5 timestamped "
                       "segments for every stream; ragged legs: U[5, 30] s clips, 2-9 segments and 40-160 tokens drawn per stream"}
    bars = {"dt_word_s": 0.02, "dconfidence": 1e-4, "dmean_logprob": 2e-4, "words_allowed_beyond": 0}
    cpu_checks = []                  # (label, audio, windows, options, words of the GPU run): replayed on the CPU path below

    # ---- one stream (the reference's shape of the call)
    def one(clip, windows=None, on=None, **kw):
        set_script(Script(windows if windows is not None else [window]))
        try:
            return wt.transcribe(on if on is not None else model, clip, language="en", fp16=False, **kw)
        finally:
            set_script(None)
    one(clips[0])                                           # warm-up: allocations, GEMM plans, the library's arenas
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    singles = [one(c) for c in clips]
    torch.cuda.synchronize()
    el1 = time.perf_counter() - t0
    n_words = sum(len(words_of(r)) for r in singles)
    assert n_words > 0
    out["1_stream"] = {"audio_s_per_s": round(30.0 * len(clips) / el1, 1), "clips": len(clips), "seconds": round(el1, 3),
                       "ms_per_clip": round(1e3 * el1 / len(clips), 1), "words": n_words}
    emit(out)

    # ---- B streams per decoder op
    def batch_of(audios, window_lists, max_streams, on=None, **kw):
        scripts = [Script(ws) for ws in window_lists]

        def on_group(idx):
            for i in idx:
                scripts[i].begin_window()
            set_row_scripts([scripts[i] for i in idx])
        streams.ON_GROUP_DECODE = on_group
        try:
            return wt.transcribe_batch(on if on is not None else model, audios, max_streams=max_streams, language="en", fp16=False, **kw)
        finally:
            streams.ON_GROUP_DECODE = None
            set_row_scripts(None)

    def many(n):
        return batch_of([clips[k % len(clips)] for k in range(n)], [[window]] * n, n)

    def histogram(sizes):
        h = {}
        for x in sizes:
            h[int(x)] = h.get(int(x), 0) + 1
        return {str(k): h[k] for k in sorted(h)}

    def driver_stats():
        d = dict(streams.LAST_RUN)
        sizes = d.pop("streams_per_loop", [])
        d["streams_per_loop_histogram"] = histogram(sizes)
        d["mean_streams_per_loop"] = round(float(np.mean(sizes)), 2) if sizes else None
        return d

    for n_streams in (B, 4 * B):
        many(n_streams)                                     # warm-up at the timed shape
        torch.cuda.synchronize()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            batch = many(n_streams)
        torch.cuda.synchronize()
        elB = (time.perf_counter() - t0) / reps
        worst = NO_GAPS
        for k, r in enumerate(batch):
            worst = merge_gaps(worst, word_gaps(words_of(r), words_of(singles[k % len(clips)]), "B-stream vs one-stream"))
        key = f"{n_streams}_streams"
        out[key] = {"audio_s_per_s": round(30.0 * n_streams / elB, 1), "clips": n_streams, "seconds": round(elB, 3),
                    "ms_per_clip": round(1e3 * elB / n_streams, 2), "words": sum(len(words_of(r)) for r in batch),
                    "speedup_vs_1_stream": round((30.0 * n_streams / elB) / (30.0 * len(clips) / el1), 2),
                    "driver": driver_stats(), "parity_vs_1_stream": gaps_report(worst)}
        out[key]["parity_vs_1_stream"]["ok"] = parity_flag(gaps_ok(worst), key, out[key]["parity_vs_1_stream"])
        if n_streams == B:
            cpu_checks.extend((f"{key}[{k}]", clips[k % len(clips)], [window], {}, words_of(batch[k])) for k in (0, 1))
        emit(out)

    # ---- the same on RAGGED work: clip lengths U[5, 30] s, a different transcript per stream
    def ragged_jobs(n, seed, make_window=H.peaked_window):
        rs = np.random.RandomState(seed)
        audios, wins, secs = [], [], []
        for k in range(n):
            sec = float(rs.uniform(5.0, 30.0))
            audios.append(clips[k % len(clips)][:int(sec * 16000)].clone())
            wins.append([make_window(rs, int(sec * 50), TS0, EOT)])
            secs.append(sec)
        return audios, wins, secs

    def ragged_parity(batch, audios, wins, n_streams, on=None):
        worst, checked = NO_GAPS, 0
        for k in range(0, n_streams, max(1, n_streams // 16)):          # 16 of the recordings, one stream at a time
            worst = merge_gaps(worst, word_gaps(words_of(batch[k]), words_of(one(audios[k], wins[k], on=on)), "ragged B-stream vs one-stream"))
            checked += 1
        return worst, checked

    for n_streams in (B, 4 * B):
        log(f"default strategy: ragged, {n_streams} streams")
        audios, wins, secs = ragged_jobs(n_streams, 100 + n_streams)
        batch_of(audios, wins, n_streams)                   # warm-up at the timed shape
        torch.cuda.synchronize()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            batch = batch_of(audios, wins, n_streams)
        torch.cuda.synchronize()
        elR = (time.perf_counter() - t0) / reps
        stats = driver_stats()
        worst, checked = ragged_parity(batch, audios, wins, n_streams)
        key = f"ragged_{n_streams}_streams"
        tok = [len(w_[0]) for w_ in wins]
        out[key] = {"audio_s_per_s": round(sum(secs) / elR, 1), "clips": n_streams, "audio_seconds": round(sum(secs), 1),
                    "clip_seconds": "U[5, 30]", "tokens_per_transcript": {"min": min(tok), "mean": round(float(np.mean(tok)), 1), "max": max(tok)},
                    "seconds": round(elR, 3), "words": sum(len(words_of(r)) for r in batch), "driver": stats,
                    "parity_vs_1_stream": gaps_report(worst, {"recordings_compared": checked})}
        out[key]["parity_vs_1_stream"]["ok"] = parity_flag(gaps_ok(worst), key, out[key]["parity_vs_1_stream"])
        # recordings of the batch that was just timed, for the CPU reference path below (8 of 32, 4 of 128)
        picks = range(0, n_streams, n_streams // 8) if n_streams == B else range(1, n_streams, n_streams // 4)
        cpu_checks.extend((f"{key}[{k}]", audios[k], wins[k], {}, words_of(batch[k])) for k in picks)
        emit(out)

    # ---- BASELINE configs[3] at N = 1: ONE long recording (1 h), its speech islands given (the reference's vad=[...] form;
    #      silero itself needs network), every island an independent unit -> the rank's islands as decoder streams.  On N
    #      ranks the same call deals the islands to the ranks first (sharding.transcribe_islands, no data-path collective).
    from whisper_timestamped.sharding import transcribe_islands
    pattern = (90, 30, 30, 60, 30, 60)                      # island lengths in seconds: one to three 30 s windows each
    n_islands = getattr(args, "e2e_islands", 72)                         # 72: 12 x 300 s = one hour
    durations = [pattern[k % len(pattern)] for k in range(n_islands)]
    total_s = sum(durations)
    assert total_s == 3600 or n_islands != 72
    hour = torch.cat([clips[k % len(clips)] for k in range(total_s // 30)])
    islands, t = [], 0.0
    for d_ in durations:
        islands.append((t, t + d_))
        t += d_
    ragged_island_windows = H.ragged_island_windows(durations, seed=77, ts0=TS0, eot=EOT, peaked=True)
    uniform_island_windows = [[window] * (d_ // 30) for d_ in durations]
    n_windows = sum(d_ // 30 for d_ in durations)
    out["long_form_1h_islands"] = {
        "islands": len(islands), "island_seconds": "30 / 60 / 90 (one to three windows each)", "windows": n_windows,
        "streams_per_decoder_op": B,
        "note": "BASELINE configs[3] at N = 1: explicit speech islands of one 1 h recording (sharding.transcribe_islands("
                "streams=B)): an island that is finished hands its place to the next one; on N ranks the islands are dealt to "
                "the ranks first.  Streams share a decoder loop only when their prompts have the same LENGTH (the decoder has no "
                "padding mask: padding would move the positions and change the result): with the reference's default "
                "condition_on_previous_text=True the later windows of a recording form their own loops until the prompt "
                "saturates at 223 tokens -- `uniform` scripts one transcript for every window (equal prompt lengths at equal "
                "window index: the best case), `ragged` draws every window's transcript (2-9 segments, 40-160 tokens) per island"}

    def island_run(window_lists, cond):
        def on_batch(indices):
            scripts = [Script(window_lists[i]) for i in indices]

            def on_group(rows):                              # rows: positions in the rank's list of islands
                for r in rows:
                    scripts[r].begin_window()
                set_row_scripts([scripts[r] for r in rows])
            streams.ON_GROUP_DECODE = on_group
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        try:
            merged = transcribe_islands(model, hour, islands, streams=B, on_batch=on_batch, language="en", fp16=False,
                                        condition_on_previous_text=cond)
        finally:
            streams.ON_GROUP_DECODE = None
            set_row_scripts(None)
        torch.cuda.synchronize()
        return merged, time.perf_counter() - t0

    def island_words(merged, i):
        s_, e_ = islands[i]
        return [(w["text"], w["start"] - s_, w["end"] - s_, w["confidence"]) for seg in merged["segments"]
                if s_ - 1e-6 <= seg["start"] < e_ - 1e-6 for w in seg["words"]]

    def island_crop(i):
        s_, e_ = islands[i]
        return hour[int(round(s_ * 16000)):int(round(e_ * 16000))]

    def island_parity(merged, window_lists, cond, picks):
        """`picks` islands: transcribe() of the island's crop, one stream, against the island's words in the merged result."""
        worst = NO_GAPS
        for i in picks:
            alone = one(island_crop(i), window_lists[i], condition_on_previous_text=cond)
            worst = merge_gaps(worst, word_gaps(island_words(merged, i), words_of(alone), f"island {i} vs transcribe(crop)"))
        rep = gaps_report(worst, {"islands_compared_with_transcribe_of_the_crop": list(picks)})
        rep["ok"] = parity_flag(gaps_ok(worst), "long_form_1h_islands", rep)
        return rep

    legs = [("condition_on_previous_text", uniform_island_windows, True), ("no_condition", uniform_island_windows, False),
            ("ragged", ragged_island_windows, True), ("ragged_no_condition", ragged_island_windows, False)]
    for label, window_lists, cond in legs:
        log(f"default strategy, long form: {label}")
        merged, el_h = island_run(window_lists, cond)
        stats = driver_stats()
        n_seg_expected = sum(sum(1 for t_ in w_[:-1] if t_ is not None and t_ >= TS0) // 2 for ws in window_lists for w_ in ws)
        assert len(merged["segments"]) == n_seg_expected, (label, len(merged["segments"]), n_seg_expected)
        starts = [s_["start"] for s_ in merged["segments"]]
        assert starts == sorted(starts) and all(len(s_["words"]) > 0 for s_ in merged["segments"])
        rec = {"audio_s_per_s": round(total_s / el_h, 1), "seconds": round(el_h, 3), "driver": stats, "segments": len(merged["segments"]),
               "words": sum(len(s_["words"]) for s_ in merged["segments"]), "condition_on_previous_text": cond}
        if label in ("condition_on_previous_text", "ragged", "ragged_no_condition"):
            rec["parity_vs_1_stream"] = island_parity(merged, window_lists, cond, [i for i in (0, 1, 3, 6, 12) if i < n_islands])
        if label in ("ragged", "ragged_no_condition"):       # islands of the timed job for the CPU reference path (30 s + 60 s)
            cpu_checks.extend((f"long_form.{label}[island {i}]", island_crop(i), window_lists[i], dict(condition_on_previous_text=cond),
                               island_words(merged, i)) for i in ((1, 3) if label == "ragged" else (2,)) if i < n_islands)
        out["long_form_1h_islands"][label] = rec
        emit(out)

    # ---- plain random-init weights (FLAT attention), the round-5 comparison: reported, gates nothing
    if dev != "cpu":
        log("default strategy: flat attention, ragged, B streams vs one stream (informational)")
        flat = load(dev, attention="flat")
        audios, wins, secs = ragged_jobs(B, 100 + B, make_window=H.ragged_window)
        batch = batch_of(audios, wins, B, on=flat)
        worst, checked = ragged_parity(batch, audios, wins, B, on=flat)
        out["flat_attention_ragged_streams_vs_1_stream"] = gaps_report(worst, {
            "recordings_compared": checked, "gates": "nothing (informational)",
            "within_the_round_5_rule_of_at_most_1_percent_of_the_words": bool(gaps_ok_between_batch_sizes(worst)),
            "why": "flat attention rows leave the DTW near-ties where a script repeats a token; the backend's GEMMs round "
                   "differently with 32 rows than with 1 (profiles/r5c_diag_ragged_parity.txt)"})
        del flat
        emit(out)

    # ---- the reference-shaped CPU path (bounded sample): two uniform clips at two thread counts (the baseline), then the
    #      recordings / islands of the TIMED batches collected above
    if not args.no_cpu_baseline:
        import cpu_kernel_standin
        from whisper_timestamped import efficient
        saved = {k: getattr(efficient, k) for k in ("REUSE_DECODER_LOGITS", "DEFER_ALIGNMENT", "GPU_FRONT_END", "FUSED_ATTENTION")}
        patch = H._Undo()
        per_check, worst_uniform, worst_timed = [], NO_GAPS, NO_GAPS
        try:
            cpu_kernel_standin.install(patch)              # kernels -> oracle/, unfused attention, backend's own log-mel
            efficient.REUSE_DECODER_LOGITS = False         # a second projection + filters per token (T.py:871-874)
            efficient.DEFER_ALIGNMENT = False              # one synchronous alignment per segment (T.py:544-557)
            model_cpu = load("cpu", attention="peaked")
            all_threads = torch.get_num_threads()
            runs = []
            # token-by-token decoding is a chain of small GEMVs: all cores of the box are not the fastest setting, so
            # the baseline is taken at the better of two thread counts (both reported)
            for k, threads in enumerate((min(16, all_threads), all_threads)):
                torch.set_num_threads(threads)
                t0 = time.perf_counter()
                try:
                    r = one(clips[k], on=model_cpu)
                finally:
                    torch.set_num_threads(all_threads)
                runs.append({"threads": threads, "seconds_per_clip": round(time.perf_counter() - t0, 2)})
                worst_uniform = merge_gaps(worst_uniform, word_gaps(words_of(singles[k]), words_of(r), "GPU one stream vs CPU path"))
                if threads == all_threads:
                    break
            best = min(runs, key=lambda x: x["seconds_per_clip"])
            torch.set_num_threads(best["threads"])
            budget = getattr(args, "e2e_cpu_parity_budget", 90.0)
            t_start = time.perf_counter()
            try:
                for label, audio, wins_, kw, gpu_words in cpu_checks:
                    if time.perf_counter() - t_start > budget:
                        per_check.append({"what": label, "skipped": "CPU budget spent"})
                        continue
                    r = one(audio, wins_, on=model_cpu, **kw)
                    gap = word_gaps(gpu_words, words_of(r), f"{label} vs the CPU path")
                    worst_timed = merge_gaps(worst_timed, gap)
                    per_check.append({"what": label, "seconds_of_audio": round(audio.numel() / 16000.0, 1), "words": gap[3],
                                      "max_abs_dt_word_s": round(gap[0], 4), "words_beyond_0.02_s": gap[4]})
            finally:
                torch.set_num_threads(all_threads)
        finally:
            patch.undo()
            for k, v in saved.items():
                setattr(efficient, k, v)
        out["cpu_baseline"] = {"value": round(30.0 / best["seconds_per_clip"], 2), "unit": "audio-seconds/s", "cores": best["threads"],
                               "kind": "port", "runs": runs,
                               "sample": "one 30 s clip per thread setting (the faster one is the baseline), one stream: the same "
                                         "whisper-base on the CPU, unfused attention with per-token QK capture, second projection "
                                         "+ logit filters per token, one alignment per segment through oracle/"}
        both = merge_gaps(worst_uniform, worst_timed)
        out["parity_vs_cpu_reference_path"] = gaps_report(both, {
            "bars": bars, "one_stream_uniform_clips": gaps_report(worst_uniform, {"clips": len(runs)}),
            "recordings_of_the_timed_batches": gaps_report(worst_timed, {"compared": sum(1 for c in per_check if "skipped" not in c),
                                                                        "each": per_check})})
        out["parity_vs_cpu_reference_path"]["ok"] = parity_flag(gaps_ok(both), "default_strategy vs the CPU reference path",
                                                                {k: v for k, v in out["parity_vs_cpu_reference_path"].items() if k != "recordings_of_the_timed_batches"})
        out["speedup_vs_cpu"] = {k: round(out[k]["audio_s_per_s"] / out["cpu_baseline"]["value"], 1)
                                 for k in ("1_stream", f"{B}_streams", f"{4 * B}_streams", f"ragged_{B}_streams", f"ragged_{4 * B}_streams")}
        emit(out)
    out["parity_failures"] = list(PARITY_FAILURES)           # [] = every parity check of this leg held
    emit(out)
    return out
