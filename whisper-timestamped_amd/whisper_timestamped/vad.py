"""Voice-activity plumbing around the transcription.

/root/reference/whisper_timestamped/transcribe.py:1870-1916 (check_vad_method), :1944-1947 + :2059-2083
(explicit-timestamp branch of get_vad_segments, dilation/merge, unit conversion), :2085-2156
(remove_non_speech: glue the speech islands together), :2158-2200 (do_convert_timestamps: map times of the
glued audio back to the original).  The third-party detectors themselves (silero, auditok: network /
onnxruntime) are outside the accelerated path: only explicit (start, end) lists are accepted here.
Speech islands are also the unit by which long-form audio is sharded across GPUs (sharding.py).
"""
from __future__ import annotations

import torch

from .words import SAMPLE_RATE


def check_vad_method(method, with_version=False):
    if method in [None, False, "False", "false", "None", "none"]:
        return None
    if method in [True, "True", "true"] or (isinstance(method, str) and (method.startswith("silero") or method == "auditok")):
        raise NotImplementedError("vad: the silero / auditok detectors are outside the accelerated path "
                                  "(SURVEY.md 8(f) N3); pass explicit (start, end) pairs in seconds")
    if not isinstance(method, str) and hasattr(method, "__iter__"):
        pairs = []
        for s_e in method:
            assert len(s_e) == 2, f"Got unexpected element {s_e} in the list of VAD segments. Expect (start, end) pairs"
            pairs.append(tuple(s_e))
        return pairs
    try:
        method = eval(method)
        assert hasattr(method, "__iter__")
    except Exception:
        raise ValueError(f"Got unexpected VAD method {method}")
    return check_vad_method(method, with_version=with_version)


def get_vad_segments(audio, sample_rate=SAMPLE_RATE, output_sample=False, min_speech_duration=0.1,
                     min_silence_duration=0.1, dilatation=0.5, method=None):
    """Explicit timestamps -> [{"start","end"}] in seconds (or samples)."""
    if not isinstance(method, list):
        raise NotImplementedError("only explicit (start, end) lists are supported")
    segments = [{"start": s * sample_rate, "end": e * sample_rate} for (s, e) in method]
    dilatation = 0                                   # explicit timestamps are taken as they are
    if dilatation > 0:                               # (kept for parity with the detectors' branch)
        pad = round(dilatation * sample_rate)
        merged = []
        for seg in segments:
            new = {"start": max(0, seg["start"] - pad), "end": min(len(audio), seg["end"] + pad)}
            if merged and merged[-1]["end"] >= new["start"]:
                merged[-1]["end"] = new["end"]
            else:
                merged.append(new)
        segments = merged
    ratio = 1 if output_sample else 1 / sample_rate
    if ratio != 1:
        for seg in segments:
            seg["start"] *= ratio
            seg["end"] *= ratio
    if output_sample:
        for seg in segments:
            seg["start"] = round(seg["start"])
            seg["end"] = round(seg["end"])
    return segments


def do_convert_timestamps(segments, t, t2=None):
    """Time(s) on the glued speech-only audio -> time(s) on the original audio.  ``segments``: (start, end) of the
    speech islands in the original; with ``t2`` both ends are kept inside the same island when possible."""
    assert len(segments)
    in_offset = out_offset = 0
    previous_end = 0
    candidates = []
    for istart, iend in segments:
        ostart = out_offset
        oend = ostart + (iend - istart)
        out_offset = oend
        in_offset += istart - previous_end
        previous_end = iend
        t_in = t <= oend
        t2_in = t_in if t2 is None else t2 <= oend
        if t_in or t2_in:
            candidates.append([max(istart, min(iend, in_offset + t)),
                               max(istart, min(iend, in_offset + t2)) if t2 is not None else None])
            if t_in and t2_in:
                break
    if not candidates:
        candidates.append([in_offset + t, in_offset + t2 if t2 is not None else None])
    if len(candidates) > 1:                          # prefer the island that preserves the duration best
        candidates = sorted(candidates, key=lambda x: abs(abs(t2 - t) - abs(x[1] - x[0])))
    best = candidates[0]
    if t2 is None:
        return round(best[0], 2)
    return [round(x, 2) for x in best]


def remove_non_speech(audio, use_sample=False, min_speech_duration=0.1, min_silence_duration=1, dilatation=0.5,
                      sample_rate=SAMPLE_RATE, method=None, avoid_empty_speech=False, plot=False):
    """-> (speech-only audio, islands [(start, end)], convert(t, t2=None))."""
    segments = get_vad_segments(audio, sample_rate=sample_rate, output_sample=True,
                                min_speech_duration=min_speech_duration, min_silence_duration=min_silence_duration,
                                dilatation=dilatation, method=method)
    segments = [(seg["start"], seg["end"]) for seg in segments]
    if len(segments) == 0:
        if avoid_empty_speech:
            segments = [(0, audio.shape[-1])]
        else:
            return torch.Tensor([]), [], lambda t, t2=None: t if t2 is None else [t, t2]
    audio_speech = torch.cat([audio[..., s:e] for s, e in segments], dim=-1)
    if plot:                                  # transcribe.py:2139-2150
        from . import plotting
        plotting.vad_figure(audio, segments, sample_rate, plot)
    if not use_sample:
        segments = [(float(s) / sample_rate, float(e) / sample_rate) for s, e in segments]
    return audio_speech, segments, lambda t, t2=None: do_convert_timestamps(segments, t, t2)
