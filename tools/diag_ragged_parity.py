#!/usr/bin/env python3
"""Where do a ragged B-stream run and the one-stream runs of the same recordings differ in word TIMES?  (bench.py's
ragged_32_streams leg, GPU call r5b: max |dt| 0.22 s on one word, confidences identical to 1e-5.)  Runs the same 32
recordings (a) twice through transcribe_batch(max_streams=32): is the path deterministic?  (b) through
transcribe_batch(max_streams=1): the streams driver with the one-stream GEMM shapes, (c) one stream at a time through
transcribe() -- and prints every word whose times differ between (a) and (c), with the (b) value beside it."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-timestamped_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import wordgaps  # noqa: E402
import many_helper as H  # noqa: E402
import whisper_double as W  # noqa: E402
from whisper_double.decoding import Script, set_row_scripts, set_script  # noqa: E402

W.install()
import whisper_timestamped as wt  # noqa: E402
from whisper_timestamped import streams, words  # noqa: E402

words.RAW_CONFIDENCE = True
dev = "cuda:0"
model = H.load_base(dev)
TS0, EOT = 50364, 50257
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
g = torch.Generator().manual_seed(7)
clips = [(0.05 * torch.randn(30 * 16000, generator=g)).float() for _ in range(4)]
rs = np.random.RandomState(100 + B)
audios, wins, secs = [], [], []
for k in range(B):
    sec = float(rs.uniform(5.0, 30.0))
    audios.append(clips[k % len(clips)][:int(sec * 16000)].clone())
    wins.append([H.ragged_window(rs, int(sec * 50), TS0, EOT)])
    secs.append(sec)


def batch(max_streams):
    scripts = [Script(w) for w in wins]

    def on_group(idx):
        for i in idx:
            scripts[i].begin_window()
        set_row_scripts([scripts[i] for i in idx])
    streams.ON_GROUP_DECODE = on_group
    try:
        return wt.transcribe_batch(model, audios, max_streams=max_streams, language="en", fp16=False)
    finally:
        streams.ON_GROUP_DECODE = None
        set_row_scripts(None)


def one(k):
    set_script(Script(wins[k]))
    try:
        return wt.transcribe(model, audios[k], language="en", fp16=False)
    finally:
        set_script(None)


a1, a2 = batch(B), batch(B)
b1 = batch(1)
c = [one(k) for k in range(B)]
wo = wordgaps.words_of
print("B-stream run 1 == run 2 (times):", all([x[1:3] for x in wo(p)] == [x[1:3] for x in wo(q)] for p, q in zip(a1, a2)))
print("streams driver, one stream at a time == transcribe():", all([x[1:3] for x in wo(p)] == [x[1:3] for x in wo(q)] for p, q in zip(b1, c)))
n_words = n_diff = 0
for k in range(B):
    for i, (x, y, z) in enumerate(zip(wo(a1[k]), wo(c[k]), wo(b1[k]))):
        n_words += 1
        if x[1:3] != y[1:3]:
            n_diff += 1
            print(json.dumps({"recording": k, "seconds": round(secs[k], 2), "word": i, "text": x[0], "B_stream": x[1:3], "one_stream": y[1:3],
                              "driver_one_at_a_time": z[1:3], "dconf": abs(x[3] - y[3])}))
print(f"{n_diff} of {n_words} words differ in time")
