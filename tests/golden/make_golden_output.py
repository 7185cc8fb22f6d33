#!/usr/bin/env python3
"""Golden for the JSON / CSV surface: the REFERENCE's filtered_keys / flatten / remove_keys / write_csv applied to the
reference's own transcribe() outputs already stored in transcribe_cases.json -> tests/golden/output_surface.json.
Build container only (imports /root/reference)."""
import io
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "whisper-timestamped_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
from golden import make_golden_transcribe as G  # noqa: E402


def surface(mod, result):
    """What the module under test (the reference here, this repository in the test) makes of one result."""
    out = {"filtered": mod.filtered_keys(result)}
    words = list(mod.flatten(result["segments"], "words"))
    out["n_words"] = len(words)
    out["segments_without_words"] = [sorted(d) for d in mod.remove_keys(result["segments"], "words")]
    for name, rows, kw in (("csv_segments", result["segments"], dict(header=True)),
                           ("csv_words_tsv", words, dict(sep="\t", text_first=False, header=["s", "e", "w"],
                                                         format_timestamps=lambda t: f"{t:.3f}")),
                           ("csv_plain", result["segments"], {})):
        buf = io.StringIO()
        mod.write_csv(rows, buf, **kw)
        out[name] = buf.getvalue()
    return out


def main():
    ref = G.load_reference()
    cases = json.load(open(os.path.join(HERE, "transcribe_cases.json"), encoding="utf-8"))
    pick = ("one_window_two_segments", "language_detection", "vad_explicit_islands", "punctuation_options", "no_confidence")
    out = {c["name"]: surface(ref, c["expected"]) for c in cases if c["name"] in pick}
    with open(os.path.join(HERE, "output_surface.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, indent=0)
    print("wrote", len(out), "cases")


if __name__ == "__main__":
    main()
