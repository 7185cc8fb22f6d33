"""Host post-processors applied to the word list after alignment.

/root/reference/whisper_timestamped/transcribe.py:2202-2262 (remove_last_null_duration_words),
:2265-2295 (ensure_increasing_positions).  Python list surgery on O(words) items, as in the reference.
"""
from __future__ import annotations

import logging

from .words import round_timestamp

logger = logging.getLogger("whisper_timestamped")


def ensure_increasing_positions(segments, min_duration=0):
    """Make start/end non-decreasing over the list (in place): an overlapping start is moved to the middle of
    the overlap (pulling the previous end back) unless that would leave the previous item shorter than
    ``min_duration``; repeated until no backward edit happens, then everything is rounded to 10 ms."""
    while True:
        moved_previous = False
        last_end = 0
        for i, item in enumerate(segments):
            if item["start"] < last_end:
                assert i > 0
                middle = round_timestamp((last_end + item["start"]) / 2)
                if middle < segments[i - 1]["start"] + min_duration:
                    middle = last_end
                else:
                    segments[i - 1]["end"] = middle
                    moved_previous = True
                item["start"] = middle
            if item["end"] <= item["start"] + min_duration:
                item["end"] = item["start"] + min_duration
            last_end = item["end"]
        if not moved_previous:
            break
    last_end = 0
    for item in segments:
        item["start"] = round_timestamp(item["start"])
        item["end"] = round_timestamp(item["end"])
        assert item["start"] >= last_end, f"Got segment {item} coming before the previous finishes ({last_end} > {item['start']})"
        assert item["end"] >= item["start"], f"Got segment {item} with end < start"
        last_end = item["end"]
    return segments


def remove_last_null_duration_words(transcription, words, recompute_text=False):
    """Drop zero-duration words at the END of each 30 s window (probable hallucinations); shortens the owning
    segment's text and removes segments left without words (in place, like the reference)."""
    window_of_segment = {}
    seek, window = None, -1
    for i, segment in enumerate(transcription["segments"]):
        if segment["seek"] != seek:
            window += 1
            seek = segment["seek"]
        window_of_segment[i] = window

    current, trailing_empty = -1, False
    doomed = []
    for i in range(len(words) - 1, -1, -1):
        word = words[i]
        empty = word["start"] == word["end"]
        idx_segment = word["idx_segment"]
        if window_of_segment[idx_segment] != current:
            trailing_empty = empty
            current = window_of_segment[idx_segment]
        elif not empty:
            trailing_empty = False
        if not trailing_empty:
            continue
        doomed.append(i)
        full_word = "".join(word["tokens"])
        segment = transcription["segments"][idx_segment]
        text = segment["text"]
        if not text.endswith(full_word):            # upstream issue #62
            if text.endswith(full_word[:-1]):
                full_word = full_word[:-1]
            elif text[:-1].endswith(full_word):
                text = text[:-1]
            else:
                raise RuntimeError(f"\"{text}\" not ending with \"{full_word}\"")
        text = text[:-len(full_word)]
        if i > 0 and words[i - 1]["idx_segment"] == idx_segment:
            segment["text"] = text
        else:
            logger.debug(f"Removing empty segment {idx_segment}")
            transcription["segments"].pop(idx_segment)
            for j in range(i + 1, len(words)):
                words[j]["idx_segment"] -= 1
        recompute_text = True
    for i in doomed:
        words.pop(i)
    if recompute_text:
        transcription["text"] = "".join(s["text"] for s in transcription["segments"])
    return transcription, words
