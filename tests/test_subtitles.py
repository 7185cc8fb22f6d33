"""whisper_timestamped.make_subtitles against the REFERENCE'S OWN functions (tests/golden/subtitles.json, written by
tests/golden/make_golden_subtitles.py from /root/reference/whisper_timestamped/make_subtitles.py)."""
import io
import json
import os
import subprocess
import sys

from golden.make_golden_subtitles import random_transcript

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "subtitles.json"), encoding="utf-8"))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _segments(case):
    if "seed" in case:
        return random_transcript(case["seed"])
    stored = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "transcribe_cases.json"), encoding="utf-8"))
    return next(c for c in stored if c["name"] == case["case"])["expected"]["segments"]


def test_split_long_segments_and_writers_equal_the_reference():
    from whisper_timestamped import make_subtitles as M
    assert len(G["cases"]) > 90
    for case in G["cases"]:
        cut = M.split_long_segments(json.loads(json.dumps(_segments(case))), case["max_length"], use_space=case["use_space"])
        assert json.loads(json.dumps(cut)) == case["split"], case.get("seed", case.get("case"))
        for name, fn in (("srt", M.write_srt), ("vtt", M.write_vtt)):
            buf = io.StringIO()
            fn(cut, file=buf)
            assert buf.getvalue() == case[name]
    for x, hours, marker, want in G["format_timestamp"]:
        assert M.format_timestamp(x, always_include_hours=hours, decimal_marker=marker) == want


def test_make_subtitles_command_line(tmp_path):
    segs = random_transcript(3)
    src = tmp_path / "clip.wav.words.json"
    src.write_text(json.dumps({"text": "", "language": "en", "segments": segs}), encoding="utf-8")
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "whisper-timestamped_amd"))
    out_dir = tmp_path / "subs"
    subprocess.run([sys.executable, "-m", "whisper_timestamped.make_subtitles", str(src), str(out_dir), "--max_length", "47"],
                   check=True, env=env)
    want = next(c for c in G["cases"] if c.get("seed") == 3 and c["max_length"] == 47)
    assert (out_dir / "clip.wav.srt").read_text(encoding="utf-8") == want["srt"]
    assert (out_dir / "clip.wav.vtt").read_text(encoding="utf-8") == want["vtt"]
    one = tmp_path / "one" / "x.vtt"
    subprocess.run([sys.executable, "-m", "whisper_timestamped.make_subtitles", str(src), str(one), "--max_length", "47"],
                   check=True, env=env)
    assert one.read_text(encoding="utf-8") == want["vtt"]
