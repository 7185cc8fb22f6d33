#!/usr/bin/env python3
"""Worker processes x decoder streams on one GPU (sharding.transcribe_many(streams=N)): a B-stream decoder loop is bound by
its one Python thread, so W processes run W loops side by side.  30 s scripted clips, whisper-base shapes.
    python tools/bench_many_streams.py     -> one JSON line per (workers, streams)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-timestamped_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import many_helper as H  # noqa: E402

if __name__ == "__main__":
    from whisper_timestamped.sharding import transcribe_many
    g = torch.Generator().manual_seed(7)
    clip = (0.05 * torch.randn(30 * 16000, generator=g)).float()
    for workers, streams in ((1, 32), (2, 32), (4, 32), (2, 64), (4, 64)):
        n = workers * streams * 2
        res, seconds = transcribe_many(H.load_base, [clip] * n, workers_per_gpu=workers, devices=["cuda:0"], warmup=True,
                                       return_timing=True, streams=streams, on_batch=H.script_batch, language="en", fp16=False)
        words = sum(len(s["words"]) for r in res for s in r["segments"])
        assert words > 0 and len(res) == n
        print(json.dumps(dict(workers=workers, streams_per_worker=streams, clips=n, seconds=round(seconds, 3),
                              audio_s_per_s=round(30.0 * n / seconds, 1), words=words)), flush=True)
