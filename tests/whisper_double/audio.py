"""whisper.audio stand-in (constants, pad_or_trim, torch.stft log-mel)."""
import numpy as np
import torch
import torch.nn.functional as F

SAMPLE_RATE = 16000
N_FFT = 400
HOP_LENGTH = 160
CHUNK_LENGTH = 30
N_SAMPLES = CHUNK_LENGTH * SAMPLE_RATE
N_FRAMES = N_SAMPLES // HOP_LENGTH
N_SAMPLES_PER_TOKEN = HOP_LENGTH * 2
FRAMES_PER_SECOND = SAMPLE_RATE // HOP_LENGTH
TOKENS_PER_SECOND = SAMPLE_RATE // N_SAMPLES_PER_TOKEN


def load_audio(file: str, sr: int = SAMPLE_RATE):
    from scipy.io import wavfile
    rate, data = wavfile.read(file)
    assert rate == sr, "whisper_double.load_audio only reads 16 kHz PCM wav files (no ffmpeg offline)"
    if data.ndim > 1:
        data = data.mean(axis=1)
    if data.dtype.kind == "i":
        data = data.astype(np.float32) / float(np.iinfo(data.dtype).max + 1)
    return data.astype(np.float32)


def pad_or_trim(array, length: int = N_SAMPLES, *, axis: int = -1):
    if torch.is_tensor(array):
        if array.shape[axis] > length:
            array = array.index_select(dim=axis, index=torch.arange(length, device=array.device))
        if array.shape[axis] < length:
            pad = [(0, 0)] * array.ndim
            pad[axis] = (0, length - array.shape[axis])
            array = F.pad(array, [p for sizes in pad[::-1] for p in sizes])
    else:
        if array.shape[axis] > length:
            array = array.take(indices=range(length), axis=axis)
        if array.shape[axis] < length:
            pad = [(0, 0)] * array.ndim
            pad[axis] = (0, length - array.shape[axis])
            array = np.pad(array, pad)
    return array


_FILTERS = {}


def mel_filters(device, n_mels: int) -> torch.Tensor:
    """Slaney mel filterbank (what whisper ships as assets/mel_filters.npz), built by transformers' implementation."""
    assert n_mels in (80, 128), f"Unsupported n_mels: {n_mels}"
    if n_mels not in _FILTERS:
        from transformers.audio_utils import mel_filter_bank
        fb = mel_filter_bank(num_frequency_bins=1 + N_FFT // 2, num_mel_filters=n_mels, min_frequency=0.0,
                             max_frequency=8000.0, sampling_rate=SAMPLE_RATE, norm="slaney", mel_scale="slaney")
        _FILTERS[n_mels] = torch.from_numpy(np.ascontiguousarray(fb.T)).float()
    return _FILTERS[n_mels].to(device)


def log_mel_spectrogram(audio, n_mels: int = 80, padding: int = 0, device=None):
    if not torch.is_tensor(audio):
        if isinstance(audio, str):
            audio = load_audio(audio)
        audio = torch.from_numpy(audio)
    if device is not None:
        audio = audio.to(device)
    if padding > 0:
        audio = F.pad(audio, (0, padding))
    window = torch.hann_window(N_FFT).to(audio.device)
    stft = torch.stft(audio, N_FFT, HOP_LENGTH, window=window, return_complex=True)
    magnitudes = stft[..., :-1].abs() ** 2
    mel_spec = mel_filters(audio.device, n_mels) @ magnitudes
    log_spec = torch.clamp(mel_spec, min=1e-10).log10()
    log_spec = torch.maximum(log_spec, log_spec.max() - 8.0)
    return (log_spec + 4.0) / 4.0
