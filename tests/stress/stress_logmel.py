#!/usr/bin/env python3
"""One-off stress (GPU box): wt_logmel_batch against the oracle (torch.stft path) for odd lengths, whole-file frame
counts, 80 / 128 mel bins, zero padding after the valid samples."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "whisper-timestamped_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from oracle import align_ref as O  # noqa: E402
from whisper_timestamped import _lib, audio as A  # noqa: E402


def main():
    rng = np.random.RandomState(9)
    worst = 0.0
    for n_mels in (80, 128):
        fb = O.mel_filters_ref(n_mels)
        for n in (201, 399, 1000, 16000 * 3 + 7, 480000, 16000 * 47 + 123, 16000 * 95):
            x = (0.1 * rng.standard_normal(n)).astype(np.float32)
            t = np.arange(n) / 16000.0
            x += (0.2 * np.sin(2 * np.pi * 330 * t)).astype(np.float32) * (np.sin(2 * np.pi * 0.5 * t) > 0)
            # (a) whole waveform, n // 160 frames (what whisper's transcribe asks for, + 30 s of padding there)
            got = A.log_mel_spectrogram(torch.from_numpy(x).cuda(), n_mels=n_mels).cpu()
            ref = O.log_mel_spectrogram_ref(torch.from_numpy(x), n_mels)
            assert got.shape == ref.shape, (got.shape, ref.shape)
            err = (got - ref).abs().max().item()
            worst = max(worst, err)
            assert err < 2e-4, (n_mels, n, err)
            # (b) crop zero-padded to 3000 frames (naive strategy), when it fits
            if n <= 480000:
                pcm = torch.zeros(1, 480000)
                pcm[0, :n] = torch.from_numpy(x)
                mel, _ = _lib.logmel(pcm.cuda(), fb, torch.tensor([n], dtype=torch.int32), n_frames=3000)
                ref3 = O.pad_or_trim_ref(ref, 3000)
                err = (mel[0].cpu() - ref3).abs().max().item()
                worst = max(worst, err)
                assert err < 2e-4, (n_mels, n, "padded", err)
        print(f"n_mels={n_mels}: ok, worst |err| so far {worst:.2e}", flush=True)


if __name__ == "__main__":
    main()
