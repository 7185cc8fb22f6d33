#!/usr/bin/env python3
"""Timing of wt_logmel_batch alone on the GPU box (both launches: STFT + finalize), several batch sizes: one JSON line
per (batch, n_mels) with the median / min microseconds over many calls and a checksum of the output (two builds that
print the same checksum computed the same values).  Round 2 used it for the A/B of the workgroup-tile kernel against
the wave-autonomous variant (tools/probes/stft_mel_wave_variant.hip, profiles/r2b_logmel_ab.jsonl)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path[:0] = [ROOT, os.path.join(ROOT, "whisper-timestamped_amd")]
    import torch
    from whisper_timestamped import _lib
    from whisper_timestamped.audio import mel_filters
    dev = "cuda:0"
    L = _lib.load()
    for n_chunks, n_mels in ((32, 80), (256, 80), (1, 80), (32, 128)):
        g = torch.Generator(device=dev).manual_seed(1)
        pcm = torch.randn((n_chunks, 480000), generator=g, device=dev) * 0.1
        fb = mel_filters(dev, n_mels)
        mel = torch.empty((n_chunks, n_mels, 3000), device=dev)
        gmax = torch.empty(n_chunks, device=dev)
        st = torch.cuda.current_stream().cuda_stream

        def call():
            _lib._check(L.wt_logmel_batch(pcm.data_ptr(), n_chunks, 480000, 0, fb.data_ptr(), n_mels, 3000, mel.data_ptr(),
                                          gmax.data_ptr(), st), "wt_logmel_batch")
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        times = []
        reps = 200 if n_chunks <= 32 else 40
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                call()
            b.record()
            torch.cuda.synchronize()
            times.append(a.elapsed_time(b) / reps * 1e3)
        print(json.dumps(dict(variant=os.environ.get("WT_AB_LABEL", "stft_mel_kernel"), n_chunks=n_chunks, n_mels=n_mels, us_median=round(sorted(times)[2], 2), us_min=round(min(times), 2),
                              checksum=float(mel.double().sum()))), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for _ in range(2):
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ), check=True)
