#!/usr/bin/env python3
"""``.words.json`` -> ``.srt`` / ``.vtt`` with long segments cut into lines.

Behavioural mirror of /root/reference/whisper_timestamped/make_subtitles.py:1-156
(same functions, same command line, same output, pinned by
tests/golden/subtitles.json = the reference's own functions on random
transcripts).  Host-side text glue: nothing here touches the GPU.

A segment longer than ``max_length`` characters is cut greedily: words are
appended while the line fits; when the next word overflows, the line is closed at
the LAST word that ended with a punctuation mark if there was one (the cue then
ends with that word and the next cue starts with the following word's start),
else right before the overflowing word.
"""
import json

from .words import _punctuation


def split_long_segments(segments, max_length, use_space=True):
    out = []
    for segment in segments:
        text = segment["text"]
        if len(text) <= max_length:
            out.append(segment)
            continue
        meta = segment["words"]
        # the visible words come from the segment text (punctuation may have been stripped from the word entries)
        words = text.split() if use_space else [w["text"] for w in meta]
        if len(words) != len(meta):
            fallback = [w["text"] for w in meta]
            print(f"WARNING: {' '.join(words)} != {' '.join(fallback)}")
            words = fallback
        sep = " " if use_space else ""
        line, line_start = "", segment["start"]
        cut = None                                   # (characters kept, cue end, next cue start) after the last punctuation
        for i, (word, m) in enumerate(zip(words, meta)):
            before = line
            line = line + sep + word if line else word
            if len(line) > max_length and before:
                if cut is not None:
                    kept, end, next_start = cut
                    out.append({"text": line[:kept], "start": line_start, "end": end})
                    line, line_start = line[kept + 1:], next_start
                else:
                    out.append({"text": before, "start": line_start, "end": meta[i - 1]["end"]})
                    line, line_start = word, m["start"]
                cut = None
            if line and line[-1] in _punctuation:    # a good place to cut later
                cut = (len(line), m["end"], meta[i + 1]["start"] if i + 1 < len(meta) else None)
        if line:
            out.append({"text": line, "start": line_start, "end": segment["end"]})
    return out


def format_timestamp(seconds: float, always_include_hours: bool = False, decimal_marker: str = "."):
    assert seconds >= 0, "non-negative timestamp expected"
    ms = round(seconds * 1000.0)
    hours, ms = divmod(ms, 3_600_000)
    minutes, ms = divmod(ms, 60_000)
    secs, ms = divmod(ms, 1_000)
    head = f"{hours:02d}:" if always_include_hours or hours > 0 else ""
    return f"{head}{minutes:02d}:{secs:02d}{decimal_marker}{ms:03d}"


def _cue_text(segment):
    return segment["text"].strip().replace("-->", "->")


def write_vtt(result, file):
    print("WEBVTT\n", file=file)
    for segment in result:
        print(f"{format_timestamp(segment['start'])} --> {format_timestamp(segment['end'])}\n{_cue_text(segment)}\n",
              file=file, flush=True)


def write_srt(result, file):
    for i, segment in enumerate(result, start=1):
        a = format_timestamp(segment["start"], always_include_hours=True, decimal_marker=",")
        b = format_timestamp(segment["end"], always_include_hours=True, decimal_marker=",")
        print(f"{i}\n{a} --> {b}\n{_cue_text(segment)}\n", file=file, flush=True)


def cli():
    import argparse
    import os
    formats = ["srt", "vtt"]
    parser = argparse.ArgumentParser(
        description="Convert .word.json transcription files (output of whisper_timestamped) to srt or vtt, being able to "
                    "cut long segments", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("input", type=str, help="Input json file, or input folder")
    parser.add_argument("output", type=str, help="Output srt or vtt file, or output folder")
    parser.add_argument("--max_length", default=200, help="Maximum length of a segment in characters", type=int)
    parser.add_argument("--format", type=str, default="all", choices=formats + ["all"],
                        help="Output format (if the output is a folder, i.e. not a file with an explicit extension)")
    args = parser.parse_args()

    to_folder = os.path.isdir(args.input) or not any(args.output.endswith(e) for e in formats)
    if to_folder:
        names = [f for f in os.listdir(args.input) if f.endswith(".words.json")] if os.path.isdir(args.input) \
            else [os.path.basename(args.input)]
        exts = [args.format] if args.format != "all" else formats
        outputs = [[os.path.join(args.output, f[:-len(".words.json")] + "." + e) for e in exts] for f in names]
        inputs = [os.path.join(args.input, f) for f in names] if os.path.isdir(args.input) else [args.input]
        os.makedirs(args.output, exist_ok=True)
    else:
        inputs, outputs = [args.input], [[args.output]]
        os.makedirs(os.path.dirname(args.output) or ".", exist_ok=True)

    for path, outs in zip(inputs, outputs):
        with open(path, "r", encoding="utf-8") as f:
            transcript = json.load(f)
        segments = transcript["segments"]
        if args.max_length:
            use_space = transcript["language"] not in ["zh", "ja", "th", "lo", "my"]
            segments = split_long_segments(segments, args.max_length, use_space=use_space)
        for out in outs:
            if out.endswith(".srt"):
                writer = write_srt
            elif out.endswith(".vtt"):
                writer = write_vtt
            else:
                raise RuntimeError(f"Unknown output format for {out}")
            with open(out, "w", encoding="utf-8") as f:
                writer(segments, file=f)


if __name__ == "__main__":
    cli()
