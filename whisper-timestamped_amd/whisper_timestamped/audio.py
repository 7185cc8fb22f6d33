"""Audio front end on the MI355X: log-mel spectrogram through libwtalign's
20x20-factored STFT kernel (no torch.stft, no rocFFT).

Counterpart of openai-whisper's audio.py as used by the reference
(/root/reference/whisper_timestamped/transcribe.py:44-47 constants,
:1213-1214 log_mel_spectrogram + pad_or_trim, :1340-1352 get_audio_tensor /
audio_minimum_padding).
"""
from __future__ import annotations

import functools

import numpy as np
import torch

from . import _lib

SAMPLE_RATE = 16000
N_FFT = 400
HOP_LENGTH = 160
CHUNK_LENGTH = 30
N_SAMPLES = CHUNK_LENGTH * SAMPLE_RATE      # 480000
N_FRAMES = N_SAMPLES // HOP_LENGTH          # 3000


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f * 3.0 / 200.0
    log_region = f >= 1000.0
    with np.errstate(divide="ignore", invalid="ignore"):
        logpart = 15.0 + 27.0 * np.log(np.where(log_region, f, 1000.0) / 1000.0) / np.log(6.4)
    return np.where(log_region, logpart, lin)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    log_region = m >= 15.0
    return np.where(log_region, 1000.0 * np.exp(np.log(6.4) / 27.0 * (m - 15.0)), m * 200.0 / 3.0)


@functools.lru_cache(maxsize=None)
def _mel_filters_np(n_mels: int) -> np.ndarray:
    """Slaney-scale, area-normalised triangular filterbank (n_mels, 201) for
    16 kHz / n_fft 400 (what whisper ships as assets/mel_filters.npz)."""
    bins = np.linspace(0.0, SAMPLE_RATE / 2, N_FFT // 2 + 1)
    edges = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(SAMPLE_RATE / 2), n_mels + 2))
    width = np.diff(edges)
    rise = (bins[None, :] - edges[:-2, None]) / width[:-1, None]
    fall = (edges[2:, None] - bins[None, :]) / width[1:, None]
    tri = np.clip(np.minimum(rise, fall), 0.0, None)
    tri *= (2.0 / (edges[2:] - edges[:-2]))[:, None]
    return tri.astype(np.float32)


_FB_ON_DEVICE = {}


def mel_filters(device, n_mels: int = 80) -> torch.Tensor:
    """One resident copy per (device, n_mels): libwtalign caches the banded form per filterbank pointer."""
    assert n_mels in (80, 128), f"Unsupported n_mels: {n_mels}"
    key = (str(torch.device(device)), n_mels)
    if key not in _FB_ON_DEVICE:
        _FB_ON_DEVICE[key] = torch.from_numpy(_mel_filters_np(n_mels)).to(device)
    return _FB_ON_DEVICE[key]


def pad_or_trim(array, length: int = N_SAMPLES, *, axis: int = -1):
    """Zero-pad or cut `array` to `length` along `axis` (torch or numpy)."""
    if torch.is_tensor(array):
        if array.shape[axis] > length:
            array = array.index_select(dim=axis, index=torch.arange(length, device=array.device))
        if array.shape[axis] < length:
            widths = [(0, 0)] * array.ndim
            widths[axis] = (0, length - array.shape[axis])
            array = torch.nn.functional.pad(array, [p for pair in widths[::-1] for p in pair])
        return array
    if array.shape[axis] > length:
        array = array.take(indices=range(length), axis=axis)
    if array.shape[axis] < length:
        widths = [(0, 0)] * array.ndim
        widths[axis] = (0, length - array.shape[axis])
        array = np.pad(array, widths)
    return array


def log_mel_spectrogram(audio, n_mels: int = 80, padding: int = 0, device=None) -> torch.Tensor:
    """(n_mels, n_samples // 160) log-mel of ONE waveform, computed on the GPU."""
    if not torch.is_tensor(audio):
        audio = torch.from_numpy(np.asarray(audio, dtype=np.float32))
    if device is None:
        device = audio.device if audio.is_cuda else torch.device("cuda", torch.cuda.current_device())
    audio = audio.to(device=device, dtype=torch.float32).reshape(1, -1)
    if padding > 0:
        audio = torch.nn.functional.pad(audio, (0, padding))
    n_frames = audio.shape[-1] // HOP_LENGTH
    mel, _ = _lib.logmel(audio, mel_filters(device, n_mels), None, n_frames=n_frames)
    return mel[0]


def log_mel_batch(pcm: torch.Tensor, n_valid_samples=None, n_mels: int = 80, n_frames: int = N_FRAMES, with_padding: bool = False,
                  launch=None):
    """(B, n_samples) crops -> (B, n_mels, n_frames), each crop normalised by its
    own max and zero-padded like log_mel_spectrogram + pad_or_trim per crop.  ``with_padding``: also find_start_padding
    of every window (int32[B] on the device, -1 = None), from a one-wave-per-window pass behind it (wt_logmel_pad_batch)."""
    if with_padding:
        mel, _, pad = _lib.logmel(pcm, mel_filters(pcm.device, n_mels), n_valid_samples, n_frames=n_frames, with_padding=True,
                                  launch=launch)
        return mel, pad
    mel, _ = _lib.logmel(pcm, mel_filters(pcm.device, n_mels), n_valid_samples, n_frames=n_frames)
    return mel


def load_audio(path: str, sr: int = SAMPLE_RATE) -> np.ndarray:
    """16 kHz mono float32 waveform.  The reference shells out to ffmpeg
    (absent here): PCM .wav files are read directly, anything else needs ffmpeg."""
    if str(path).lower().endswith(".wav"):
        from scipy.io import wavfile
        rate, data = wavfile.read(path)
        if data.ndim > 1:
            data = data.mean(axis=1)
        if data.dtype.kind == "i":
            data = data.astype(np.float32) / float(np.iinfo(data.dtype).max + 1)
        data = data.astype(np.float32)
        if rate != sr:
            from scipy.signal import resample_poly
            g = np.gcd(rate, sr)
            data = resample_poly(data, sr // g, rate // g).astype(np.float32)
        return data
    import shutil
    import subprocess
    if shutil.which("ffmpeg") is None:
        raise RuntimeError(f"ffmpeg is needed to decode {path!r} and is not installed")
    cmd = ["ffmpeg", "-nostdin", "-threads", "0", "-i", path, "-f", "s16le", "-ac", "1", "-acodec", "pcm_s16le",
           "-ar", str(sr), "-"]
    out = subprocess.run(cmd, capture_output=True, check=True).stdout
    return np.frombuffer(out, np.int16).flatten().astype(np.float32) / 32768.0
