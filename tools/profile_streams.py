#!/usr/bin/env python3
"""Where does the B-stream default strategy (whisper_timestamped/streams.py) spend its wall time?  cProfile of
transcribe_batch on B synthetic 30 s clips with the bench's scripted transcript (whisper double, whisper-base shapes).
    python tools/profile_streams.py [B]        (GPU box; prints the top cumulative entries)"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-timestamped_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import many_helper as H  # noqa: E402
import whisper_double as W  # noqa: E402
from whisper_double.decoding import Script, set_row_scripts  # noqa: E402
from golden import make_golden_transcribe as G  # noqa: E402

W.install()
import whisper_timestamped as wt  # noqa: E402
from whisper_timestamped import streams  # noqa: E402

if os.environ.get("WT_PAUSE_GC") == "0":       # A/B: the cyclic collector left on while the batch decodes
    streams.PAUSE_GC = False
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
model = H.load_base("cuda:0")
g = torch.Generator().manual_seed(7)
clips = [(0.05 * torch.randn(30 * 16000, generator=g)).float() for _ in range(4)]
window = G.window_script(50364, 50257, [(s, [None] * n, e) for s, n, e in H.SEGMENTS], "eot")


def many(n):
    scripts = [Script([window]) for _ in range(n)]

    def on_group(idx):
        for i in idx:
            scripts[i].begin_window()
        set_row_scripts([scripts[i] for i in idx])
    streams.ON_GROUP_DECODE = on_group
    try:
        return wt.transcribe_batch(model, [clips[k % 4] for k in range(n)], max_streams=n, language="en", fp16=False)
    finally:
        streams.ON_GROUP_DECODE = None
        set_row_scripts(None)


many(B)
torch.cuda.synchronize()
t0 = time.perf_counter()
many(B)
torch.cuda.synchronize()
print(f"B={B}: {time.perf_counter() - t0:.3f} s per batch = {30 * B / (time.perf_counter() - t0):.0f} audio-s/s")
pr = cProfile.Profile()
pr.enable()
many(B)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(25)
for fn in ("_finish_group", "replay_window", "_align_open_segment", "_close_window", "take_result", "launch", "_collect", "commit"):
    st.sort_stats("cumulative").print_callees(fn)
