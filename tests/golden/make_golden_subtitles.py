#!/usr/bin/env python3
"""tests/golden/subtitles.json: the REFERENCE's make_subtitles.py (split_long_segments, write_srt, write_vtt,
format_timestamp -- /root/reference/whisper_timestamped/make_subtitles.py, imported unmodified: it has no third-party
dependency) on random word-level transcripts and on reference outputs already stored in transcribe_cases.json.
Build container only."""
import importlib.util
import io
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/whisper_timestamped/make_subtitles.py"


def random_transcript(seed):
    rng = np.random.RandomState(seed)
    syll = ["ka", "to", "mi", "ra", "ne", "so", "lu", "vi", "pa", "de"]
    punct = [",", ".", "?", "!", "…", ";"]
    segments, t = [], 0.0
    for _ in range(int(rng.randint(1, 5))):
        words = []
        for _ in range(int(rng.randint(1, 40))):
            w = "".join(syll[rng.randint(10)] for _ in range(int(rng.randint(1, 5))))
            if rng.rand() < 0.25:
                w += punct[rng.randint(len(punct))]
            a = round(t + float(rng.rand()) * 0.2, 2)
            b = round(a + 0.05 + float(rng.rand()) * 0.5, 2)
            words.append({"text": w, "start": a, "end": b, "confidence": round(float(rng.rand()), 3)})
            t = b
        segments.append({"text": " " + " ".join(w["text"] for w in words), "start": words[0]["start"], "end": words[-1]["end"],
                         "words": words})
    return segments


def render(mod, segments):
    out = {}
    for name, fn in (("srt", mod.write_srt), ("vtt", mod.write_vtt)):
        buf = io.StringIO()
        fn(segments, file=buf)
        out[name] = buf.getvalue()
    return out


def main():
    spec = importlib.util.spec_from_file_location("ref_make_subtitles", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    cases = []
    for seed in range(24):
        segs = random_transcript(seed)
        for max_length, use_space in ((20, True), (47, True), (200, True), (15, False)):
            cut = ref.split_long_segments(json.loads(json.dumps(segs)), max_length, use_space=use_space)
            cases.append(dict(seed=seed, max_length=max_length, use_space=use_space, split=cut, **render(ref, cut)))
    stored = json.load(open(os.path.join(HERE, "transcribe_cases.json"), encoding="utf-8"))
    for c in stored:
        if c["name"] in ("one_window_two_segments", "two_windows_prompted", "punctuation_options"):
            segs = c["expected"]["segments"]
            cut = ref.split_long_segments(json.loads(json.dumps(segs)), 30, use_space=True)
            cases.append(dict(case=c["name"], max_length=30, use_space=True, split=cut, **render(ref, cut)))
    stamps = [(x, h, m, ref.format_timestamp(x, always_include_hours=h, decimal_marker=m))
              for x in (0.0, 0.0004, 1.2345, 59.9996, 61.5, 3599.999, 3600.0, 86399.5) for h in (False, True) for m in (".", ",")]
    with open(os.path.join(HERE, "subtitles.json"), "w", encoding="utf-8") as f:
        json.dump(dict(cases=cases, format_timestamp=stamps), f, ensure_ascii=False, indent=0)
    print("wrote subtitles.json:", len(cases), "cases")


if __name__ == "__main__":
    main()
