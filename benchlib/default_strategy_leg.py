"""transcribe()-level leg: the DEFAULT strategy (the reference's efficient strategy), one stream and B decoder streams,
uniform / ragged / long-form work, the reference-shaped CPU path beside it and the per-word parity checks."""
import os
import sys
import time

import numpy as np
import torch

from .common import log, make_emitter
from wordgaps import (NO_GAPS, PARITY_FAILURES, gaps_ok_between_batch_sizes, gaps_report, merge_gaps, parity_flag, word_gaps,
                      words_of)


def run_efficient_leg(args, emit):
    """The DEFAULT strategy of transcribe() (the reference's efficient strategy: word alignment on the fly while the
    backend decodes, T.py:359-1001), whisper double as the model:
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
      cpu_baseline      the reference-shaped CPU path for the same clips: the same model on the host cores, unfused
                        attention with per-token QK capture, a second projection + logit filters per token, one
                        synchronous alignment per segment through oracle/ (the reference's shape, T.py:783-793,849-881,
                        544-557), one stream -- a bounded sample;
      parity            every B-stream recording against the one-stream output (word times, raw confidences, mean
                        log-probabilities) and the sampled clips against the CPU path's."""
    import many_helper as H          # tests/: the whisper double as the model, the scripted transcript
    import whisper_double as W
    from whisper_double.decoding import Script, set_row_scripts, set_script
    from golden import make_golden_transcribe as G
    W.install()
    import whisper_timestamped as wt
    from whisper_timestamped import streams, words
    words.RAW_CONFIDENCE = True      # confidences before the reference's round(, 3): parity is asserted on the raw values
    dev = getattr(args, "e2e_device", "cuda:0")     # (a CPU dry run of this leg's host logic: tools/dry_run_efficient_leg.py)
    model = H.load_base(dev)
    B = args.e2e_streams
    TS0, EOT = 50364, 50257
    g = torch.Generator().manual_seed(7)
    clips = [(0.05 * torch.randn(30 * 16000, generator=g)).float() for _ in range(4)]
    segs = [(s, [None] * n, e) for s, n, e in H.SEGMENTS]
    window = G.window_script(TS0, EOT, segs, "eot")
    out = {"workload": "whisper-base (random init, fp32), synthetic clips, scripted transcripts, transcribe() with its defaults "
                       "(efficient strategy, greedy, condition_on_previous_text=True); uniform legs: 30 s clips, one ~110-token "
                       "transcript in 5 timestamped segments for every stream; ragged legs: U[5, 30] s clips, 2-9 segments and "
                       "40-160 tokens drawn per stream"}
    bars = {"dt_word_s": 0.02, "dconfidence": 1e-4, "dmean_logprob": 2e-4}

    # ---- one stream (the reference's shape of the call)
    def one(clip, windows=None, **kw):
        set_script(Script(windows if windows is not None else [window]))
        try:
            return wt.transcribe(model, clip, language="en", fp16=False, **kw)
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
    def batch_of(audios, window_lists, max_streams, **kw):
        scripts = [Script(ws) for ws in window_lists]

        def on_group(idx):
            for i in idx:
                scripts[i].begin_window()
            set_row_scripts([scripts[i] for i in idx])
        streams.ON_GROUP_DECODE = on_group
        try:
            return wt.transcribe_batch(model, audios, max_streams=max_streams, language="en", fp16=False, **kw)
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
        out[key]["parity_vs_1_stream"]["ok"] = parity_flag(gaps_ok_between_batch_sizes(worst), key, out[key]["parity_vs_1_stream"])
        emit(out)

    # ---- the same on RAGGED work: clip lengths U[5, 30] s, a different transcript per stream
    def ragged_jobs(n, seed):
        rs = np.random.RandomState(seed)
        audios, wins, secs = [], [], []
        for k in range(n):
            sec = float(rs.uniform(5.0, 30.0))
            audios.append(clips[k % len(clips)][:int(sec * 16000)].clone())
            wins.append([H.ragged_window(rs, int(sec * 50), TS0, EOT)])
            secs.append(sec)
        return audios, wins, secs
    for n_streams in (B, 4 * B):
        print(f"[bench] default strategy: ragged, {n_streams} streams", file=sys.stderr, flush=True)
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
        worst, checked = NO_GAPS, 0
        for k in range(0, n_streams, max(1, n_streams // 16)):          # 16 of the recordings, one stream at a time
            worst = merge_gaps(worst, word_gaps(words_of(batch[k]), words_of(one(audios[k], wins[k])), "ragged B-stream vs one-stream"))
            checked += 1
        key = f"ragged_{n_streams}_streams"
        tok = [len(w_[0]) for w_ in wins]
        out[key] = {"audio_s_per_s": round(sum(secs) / elR, 1), "clips": n_streams, "audio_seconds": round(sum(secs), 1),
                    "clip_seconds": "U[5, 30]", "tokens_per_transcript": {"min": min(tok), "mean": round(float(np.mean(tok)), 1), "max": max(tok)},
                    "seconds": round(elR, 3), "words": sum(len(words_of(r)) for r in batch), "driver": stats,
                    "parity_vs_1_stream": gaps_report(worst, {"recordings_compared": checked})}
        out[key]["parity_vs_1_stream"]["ok"] = parity_flag(gaps_ok_between_batch_sizes(worst), key, out[key]["parity_vs_1_stream"])
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
    ragged_island_windows = H.ragged_island_windows(durations, seed=77, ts0=TS0, eot=EOT)
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

    def island_run(window_lists, cond, hold=0):
        def on_batch(indices):
            scripts = [Script(window_lists[i]) for i in indices]

            def on_group(rows):                              # rows: positions in the rank's list of islands
                for r in rows:
                    scripts[r].begin_window()
                set_row_scripts([scripts[r] for r in rows])
            streams.ON_GROUP_DECODE = on_group
        streams.HOLD_FOR_BUCKET = hold
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        try:
            merged = transcribe_islands(model, hour, islands, streams=B, on_batch=on_batch, language="en", fp16=False,
                                        condition_on_previous_text=cond)
        finally:
            streams.ON_GROUP_DECODE = None
            streams.HOLD_FOR_BUCKET = 0
            set_row_scripts(None)
        torch.cuda.synchronize()
        return merged, time.perf_counter() - t0

    def island_parity(merged, window_lists, cond, picks):
        """`picks` islands: transcribe() of the island's crop, one stream, against the island's words in the merged result."""
        worst = NO_GAPS
        for i in picks:
            s_, e_ = islands[i]
            crop = hour[int(round(s_ * 16000)):int(round(e_ * 16000))]
            alone = one(crop, window_lists[i], condition_on_previous_text=cond)
            mine = [(w["text"], w["start"] - s_, w["end"] - s_, w["confidence"]) for seg in merged["segments"]
                    if s_ - 1e-6 <= seg["start"] < e_ - 1e-6 for w in seg["words"]]
            worst = merge_gaps(worst, word_gaps(mine, words_of(alone), f"island {i} vs transcribe(crop)"))
        rep = gaps_report(worst, {"islands_compared_with_transcribe_of_the_crop": list(picks)})
        rep["ok"] = parity_flag(gaps_ok_between_batch_sizes(worst), "long_form_1h_islands", rep)
        return rep

    ragged_words_per_island = []
    legs = [("condition_on_previous_text", uniform_island_windows, True, 0), ("no_condition", uniform_island_windows, False, 0),
            ("ragged", ragged_island_windows, True, 0), ("ragged_bucket_admission", ragged_island_windows, True, 2),
            ("ragged_no_condition", ragged_island_windows, False, 0)]
    for label, window_lists, cond, hold in legs:
        print(f"[bench] default strategy, long form: {label}", file=sys.stderr, flush=True)
        merged, el_h = island_run(window_lists, cond, hold)
        stats = driver_stats()
        n_seg_expected = sum(sum(1 for t_ in w_[:-1] if t_ is not None and t_ >= TS0) // 2 for ws in window_lists for w_ in ws)
        assert len(merged["segments"]) == n_seg_expected, (label, len(merged["segments"]), n_seg_expected)
        starts = [s_["start"] for s_ in merged["segments"]]
        assert starts == sorted(starts) and all(len(s_["words"]) > 0 for s_ in merged["segments"])
        rec = {"audio_s_per_s": round(total_s / el_h, 1), "seconds": round(el_h, 3), "driver": stats, "segments": len(merged["segments"]),
               "words": sum(len(s_["words"]) for s_ in merged["segments"]), "condition_on_previous_text": cond}
        if hold:
            rec["hold_for_bucket"] = hold
        if label == "ragged":                                   # (per island, for the worker-process leg below)
            per = []
            for s_, e_ in islands:
                per.append([(w["text"], round(w["start"] - s_, 2), round(w["end"] - s_, 2)) for seg in merged["segments"]
                            if s_ - 1e-6 <= seg["start"] < e_ - 1e-6 for w in seg["words"]])
            ragged_words_per_island[:] = per
        if label in ("condition_on_previous_text", "ragged"):
            rec["parity_vs_1_stream"] = island_parity(merged, window_lists, cond, [i for i in (0, 1, 3, 6, 12) if i < n_islands])
        out["long_form_1h_islands"][label] = rec
        emit(out)

    # ---- what recovers the ragged loss on ONE GPU: processes.  A decoder loop is bound by its one Python thread whether it
    #      carries 1 stream or 32, so W worker processes (own interpreter, own HIP queues, own copy of the 290 MB model) run W
    #      loops side by side: the islands as independent recordings through sharding.transcribe_many(streams=B / W).  On N
    #      GPUs the N ranks ARE such processes.  (Timed between the workers' common start and the last result; process
    #      start-up and model load are reported beside it.)
    if dev != "cpu" and getattr(args, "e2e_worker_processes", 0) > 1:
        import functools
        from whisper_timestamped.sharding import transcribe_many
        W_ = int(getattr(args, "e2e_worker_processes", 0))
        print(f"[bench] default strategy, long form: ragged, {W_} worker processes", file=sys.stderr, flush=True)
        crops = [hour[int(round(s_ * 16000)):int(round(e_ * 16000))].clone() for s_, e_ in islands]
        t0 = time.perf_counter()
        try:
            res_w, slowest = transcribe_many(H.load_base, crops, workers_per_gpu=W_, devices=[dev], warmup=True, return_timing=True,
                                             streams=max(1, B // W_), language="en", fp16=False,
                                             on_batch=functools.partial(H.script_ragged_islands, durations=tuple(durations), seed=77))
            wall = time.perf_counter() - t0
            ref, n_w, n_moved = ragged_words_per_island, 0, 0
            for i, r in enumerate(res_w):                         # (8 streams per loop there, 32 here: batch-size rounding, see word_gaps)
                mine = [(x[0], x[1], x[2]) for x in words_of(r)]
                assert [x[0] for x in mine] == [x[0] for x in ref[i]], f"island {i}: words differ between one process and {W_}"
                n_w += len(mine)
                n_moved += sum(max(abs(a_[1] - b_[1]), abs(a_[2] - b_[2])) > 0.02 + 1e-9 for a_, b_ in zip(mine, ref[i]))
            out["long_form_1h_islands"]["ragged_worker_processes"] = {
                "worker_processes": W_, "streams_per_process": max(1, B // W_), "audio_s_per_s": round(total_s / slowest, 1),
                "seconds": round(slowest, 3), "seconds_incl_process_start_and_model_load": round(wall, 2),
                "words_compared_with_the_one_process_run": n_w, "words_beyond_0.02_s": int(n_moved),
                "vs_one_process": round((total_s / slowest) / out["long_form_1h_islands"]["ragged"]["audio_s_per_s"], 2)}
        except Exception as e:                                   # noqa: BLE001 -- an optional leg must not cost the others
            out["long_form_1h_islands"]["ragged_worker_processes"] = {"error": repr(e)[:300]}
        emit(out)

    # ---- the reference-shaped CPU path, same clips (bounded sample)
    if not args.no_cpu_baseline:
        import cpu_kernel_standin
        from whisper_timestamped import efficient
        saved = {k: getattr(efficient, k) for k in ("REUSE_DECODER_LOGITS", "DEFER_ALIGNMENT", "GPU_FRONT_END", "FUSED_ATTENTION")}
        patch = H._Undo()
        try:
            cpu_kernel_standin.install(patch)              # kernels -> oracle/, unfused attention, backend's own log-mel
            efficient.REUSE_DECODER_LOGITS = False         # a second projection + filters per token (T.py:871-874)
            efficient.DEFER_ALIGNMENT = False              # one synchronous alignment per segment (T.py:544-557)
            model_cpu = H.load_base("cpu")
            all_threads = torch.get_num_threads()
            runs, worst = [], NO_GAPS
            # token-by-token decoding is a chain of small GEMVs: all cores of the box are not the fastest setting, so
            # the baseline is taken at the better of two thread counts (both reported)
            for k, threads in enumerate((min(16, all_threads), all_threads)):
                torch.set_num_threads(threads)
                set_script(Script([window]))
                t0 = time.perf_counter()
                try:
                    r = wt.transcribe(model_cpu, clips[k], language="en", fp16=False)
                finally:
                    set_script(None)
                    torch.set_num_threads(all_threads)
                runs.append({"threads": threads, "seconds_per_clip": round(time.perf_counter() - t0, 2)})
                worst = merge_gaps(worst, word_gaps(words_of(r), words_of(singles[k]), "GPU vs CPU path"))
                if threads == all_threads:
                    break
            # one RAGGED clip as well (its own transcript), at the faster thread count
            best = min(runs, key=lambda x: x["seconds_per_clip"])
            audios, wins, _ = ragged_jobs(B, 100 + B)
            torch.set_num_threads(best["threads"])
            set_script(Script(wins[1]))
            try:
                r = wt.transcribe(model_cpu, audios[1], language="en", fp16=False)
            finally:
                set_script(None)
                torch.set_num_threads(all_threads)
            gpu_same = None
        finally:
            patch.undo()
            for k, v in saved.items():
                setattr(efficient, k, v)
        gpu_same = one(audios[1], wins[1])
        ragged_gap = word_gaps(words_of(gpu_same), words_of(r), "GPU vs CPU path, ragged clip")
        out["cpu_baseline"] = {"value": round(30.0 / best["seconds_per_clip"], 2), "unit": "audio-seconds/s", "cores": best["threads"],
                               "kind": "port", "runs": runs,
                               "sample": f"one 30 s clip per thread setting (the faster one is the baseline), one stream: the same "
                                         f"whisper-base on the CPU, unfused attention with per-token QK capture, second projection "
                                         f"+ logit filters per token, one alignment per segment through oracle/"}
        out["parity_vs_cpu_reference_path"] = gaps_report(worst, {"clips": len(runs), "bars": bars})
        # (a ragged clip as well, reported on its own: the CPU's and the GPU's fp32 GEMMs round differently, and on a
        #  repeated token a random-init model's flat attention leaves the DTW near-ties -- see the note above word_gaps)
        out["parity_vs_cpu_reference_path"]["ragged_clip"] = gaps_report(ragged_gap, {"seconds": round(audios[1].numel() / 16000.0, 2)})
        out["parity_vs_cpu_reference_path"]["ok"] = parity_flag(
            worst[0] <= 0.02 + 1e-9 and worst[1] <= 1e-4 and worst[2] <= 2e-4 and ragged_gap[1] <= 1e-4 and ragged_gap[2] <= 2e-4,
            "default_strategy vs the CPU reference path", out["parity_vs_cpu_reference_path"])
        out["speedup_vs_cpu"] = {k: round(out[k]["audio_s_per_s"] / out["cpu_baseline"]["value"], 1)
                                 for k in ("1_stream", f"{B}_streams", f"{4 * B}_streams", f"ragged_{B}_streams", f"ragged_{4 * B}_streams")}
        emit(out)
    out["parity_failures"] = list(PARITY_FAILURES)           # [] = every parity check of this leg held
    emit(out)
    return out

