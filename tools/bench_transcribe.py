#!/usr/bin/env python3
"""What do word timestamps cost on top of plain decoding?  (GPU box; whisper double, random weights, scripted tokens)

Times, for one 30 s clip and a scripted ~110-token transcript on a whisper-base-shaped model:
  plain      : the backend's own transcribe() (no hooks)
  unfused    : this repository's transcribe() with efficient.FUSED_ATTENTION = False (qk observed inside
               whisper.model.disable_sdpa(), the only way the reference can get it)
  timestamped: this repository's transcribe() with its defaults (wt_qk_rows_batch per token, logits reuse verified on
               the first token of every window, one alignment launch set per window)
  reuse      : efficient.REUSE_DECODER_LOGITS = True (no verification pass at all)
  no_reuse   : efficient.REUSE_DECODER_LOGITS = False (the reference's second projection + filter pass per token)
  per_segment: timestamped with efficient.DEFER_ALIGNMENT = False (one synchronous alignment per flushed segment)
Prints one JSON line.  The decode loop itself is the backend's Python loop (batch 1), as with the reference.

usage: bench_transcribe.py [base|small|...] [--only plain|timestamped|unfused|reuse|no_reuse|per_segment]
       (--only: that variant alone, for a rocprofv3 --kernel-trace of exactly one data plane)
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-timestamped_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import whisper_double as W  # noqa: E402
from whisper_double.decoding import Script, set_script  # noqa: E402
from golden import make_golden_transcribe as G  # noqa: E402

W.install()
import whisper_timestamped as wt  # noqa: E402
from whisper_timestamped import efficient  # noqa: E402


def main():
    dev = "cuda:0"
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    if only:
        args = [a for a in args if a != only]
    name = args[0] if args else "base"
    model = W.build_model(name, seed=0, device=dev)
    g = torch.Generator().manual_seed(5)
    audio = (0.05 * torch.randn(30 * 16000, generator=g)).float()
    ML, EOT = 50364, 50257
    segs = [(s, [None] * n, e) for s, n, e in [(0, 20, 280), (300, 22, 600), (620, 18, 900), (920, 21, 1200), (1220, 19, 1490)]]
    script = [G.window_script(ML, EOT, segs, "eot")]
    n_tokens = len(script[0])

    def timed(fn, reps=3):
        best = None
        for _ in range(reps):
            set_script(Script(script))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            set_script(None)
            best = dt if best is None else min(best, dt)
        return best, out

    if only:
        def variant():
            if only == "plain":
                with torch.no_grad():
                    return model.transcribe(audio, language="en", temperature=0.0, fp16=False)
            efficient.FUSED_ATTENTION = only != "unfused"
            efficient.REUSE_DECODER_LOGITS = True if only == "reuse" else False if only in ("no_reuse", "unfused") else "auto"
            efficient.DEFER_ALIGNMENT = only != "per_segment"
            return wt.transcribe(model, audio, language="en", fp16=False)
        t, _ = timed(variant, reps=4)
        print(json.dumps(dict(model=f"whisper-{name} shapes (random init)", variant=only, tokens=n_tokens, runs=4,
                              best_s=round(t, 4), ms_per_token=round(1e3 * t / n_tokens, 3))))
        return
    with torch.no_grad():
        t_plain, _ = timed(lambda: model.transcribe(audio, language="en", temperature=0.0, fp16=False))
    efficient.FUSED_ATTENTION = False             # the reference's way: every attention module unfused, qk observed,
    efficient.REUSE_DECODER_LOGITS = False        # a second projection + filter pass per token
    t_unfused, res0 = timed(lambda: wt.transcribe(model, audio, language="en", fp16=False))
    efficient.FUSED_ATTENTION = True
    t_noreuse, res4 = timed(lambda: wt.transcribe(model, audio, language="en", fp16=False))
    efficient.REUSE_DECODER_LOGITS = "auto"       # the defaults
    t_ts, res = timed(lambda: wt.transcribe(model, audio, language="en", fp16=False))
    stats = dict(efficient.LAST_SESSION)
    efficient.REUSE_DECODER_LOGITS = True
    t_reuse, res2 = timed(lambda: wt.transcribe(model, audio, language="en", fp16=False))
    efficient.REUSE_DECODER_LOGITS = "auto"
    efficient.DEFER_ALIGNMENT = False
    t_seg, res3 = timed(lambda: wt.transcribe(model, audio, language="en", fp16=False))
    efficient.DEFER_ALIGNMENT = True
    words = sum(len(s.get("words", [])) for s in res["segments"])
    starts = lambda r: [(w["start"], w["end"]) for s in r["segments"] for w in s["words"]]  # noqa: E731
    same = starts(res) == starts(res2) == starts(res0) == starts(res3) == starts(res4)
    print(json.dumps(dict(model=f"whisper-{name} shapes (random init)", tokens=n_tokens, segments=len(res["segments"]), words=words,
                          plain_s=round(t_plain, 4), timestamped_unfused_attention_s=round(t_unfused, 4),
                          timestamped_s=round(t_ts, 4), timestamped_reuse_s=round(t_reuse, 4),
                          timestamped_reference_logits_s=round(t_noreuse, 4),
                          overhead_reference_logits_pct=round(100 * (t_noreuse / t_plain - 1), 1),
                          default_session=stats,
                          timestamped_per_segment_sync_s=round(t_seg, 4),
                          overhead_per_segment_sync_pct=round(100 * (t_seg / t_plain - 1), 1),
                          overhead_unfused_attention_pct=round(100 * (t_unfused / t_plain - 1), 1),
                          overhead_pct=round(100 * (t_ts / t_plain - 1), 1),
                          overhead_reuse_pct=round(100 * (t_reuse / t_plain - 1), 1),
                          ms_per_token_plain=round(1e3 * t_plain / n_tokens, 3), same_word_times=same)))


if __name__ == "__main__":
    main()
