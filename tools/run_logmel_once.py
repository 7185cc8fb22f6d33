#!/usr/bin/env python3
"""wt_logmel_batch on 32 x 30 s chunks, a few calls: the command rocprofv3 --pmc wraps (tools/pmc_counters.py reads
the result).  WT_LOGMEL_TILES=1 selects the workgroup-tile kernel."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "whisper-timestamped_amd")]
import torch  # noqa: E402
from whisper_timestamped import _lib  # noqa: E402
from whisper_timestamped.audio import mel_filters  # noqa: E402

dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
pcm = torch.randn((n, 480000), device=dev) * 0.1
for _ in range(6):
    mel, _ = _lib.logmel(pcm, mel_filters(dev, 80))
torch.cuda.synchronize()
print("ok", float(mel.sum()))
