"""tools/replay_reference_goldens.py: the harness that replays the reference's own end-to-end
goldens (/root/reference/tests/expected/*.words.json, tolerance of tests/test_transcribe.py:185-198) through this
repository.  Real checkpoints, openai-whisper and ffmpeg are absent from the build image, so here

  * the harness must SKIP every case and name what it needs (and still exit 0: nothing was verified, nothing failed);
  * its mechanics (run -> loose comparison -> gaps -> exit status, the dump of a differing output) are exercised on a
    fabricated "reference" tree with the whisper double as the model and the CPU oracle as the kernels;
  * the last test is the real replay: it runs wherever openai-whisper, the checkpoints and a GPU exist.
"""
import copy
import importlib.util
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_tool():
    spec = importlib.util.spec_from_file_location("replay_reference_goldens", os.path.join(ROOT, "tools", "replay_reference_goldens.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_loose_is_the_references_tolerance():
    R = load_tool()
    assert R.loose({"a": [0.449, -0.04, 1.25, (2.06, "x")], "b": 3}) == {"a": [0.4, 0.0, 1.2, (2.1, "x")], "b": 3}
    got = {"segments": [{"words": [{"text": "a", "start": 0.10, "end": 0.31, "confidence": 0.5}]}]}
    want = {"segments": [{"words": [{"text": "a", "start": 0.12, "end": 0.30, "confidence": 0.501}]}]}
    dt, dc, ok = R.word_gaps(got, want)
    assert ok and abs(dt - 0.02) < 1e-9 and abs(dc - 0.001) < 1e-9
    assert R.word_gaps(got, {"segments": [{"words": [{"text": "b", "start": 0, "end": 0}]}]})[2] is False


def test_without_checkpoints_every_case_is_skipped_and_named(tmp_path):
    R = load_tool()
    args = R.argparse.Namespace(reference=str(tmp_path), download_root=str(tmp_path), device="cuda", long=True, dump=None)
    rep = R.replay(args)
    assert rep["replayed"] == 0 and rep["failed"] == 0 and rep["skipped"] == len(rep["cases"]) == 13
    for c in rep["cases"]:
        assert c["status"] == "skipped" and any(n.startswith("checkpoint ") for n in c["needs"])
    assert R.main(["--reference", str(tmp_path), "--download_root", str(tmp_path)]) == 0


def test_mechanics_on_a_fabricated_reference_tree(tmp_path, monkeypatch):
    """One case really runs (tiny_auto/bonjour.wav: the whisper double as `load_model("tiny")`, oracle kernels); the
    "golden" is this run's own output, then a perturbed copy -- the harness must say ok, then DIFFERENT + exit 1."""
    import cpu_kernel_standin
    import whisper_double as W
    from scipy.io import wavfile
    W.install()
    cpu_kernel_standin.install(monkeypatch)
    import whisper_timestamped as wt
    R = load_tool()
    ref = tmp_path / "reference"
    (ref / "tests" / "data").mkdir(parents=True)
    (ref / "tests" / "expected" / "tiny_auto").mkdir(parents=True)
    rng = np.random.RandomState(0)
    t = np.arange(int(2.5 * 16000)) / 16000.0
    wavfile.write(str(ref / "tests" / "data" / "bonjour.wav"), 16000,
                  ((0.05 * rng.standard_normal(t.shape) + 0.1 * np.sin(2 * np.pi * 220.0 * t)) * 32767).astype(np.int16))
    monkeypatch.setattr(wt, "load_model", lambda name, device=None, download_root=None: W.build_model(name, seed=0, device="cpu"))
    real_req = R.requirements
    monkeypatch.setattr(R, "requirements", lambda a: dict(real_req(a), **{"openai-whisper": "double", "gpu": "oracle kernels",
                                                                         "checkpoint tiny": "random init"}))
    args = R.argparse.Namespace(reference=str(ref), download_root=None, device="cpu", long=False, dump=str(tmp_path / "dump"))
    case = next(c for c in R.cases(args) if c["name"] == "tiny_auto/bonjour.wav")
    first = R.run_case(case, args, {})
    assert first["segments"] and "words" in first["segments"][0]
    with open(case["expected"], "w", encoding="utf-8") as f:
        json.dump(first, f, ensure_ascii=False)
    rep = R.replay(args)
    ran = [c for c in rep["cases"] if c["status"] != "skipped"]
    assert len(ran) == 1 and ran[0]["status"] == "ok" and ran[0]["max_abs_dt_word_s"] == 0 and ran[0]["within_baseline_bars"]
    assert rep["replayed"] == 1 and rep["failed"] == 0 and rep["skipped"] == 10
    moved = copy.deepcopy(first)
    moved["segments"][0]["words"][0]["start"] += 0.3
    with open(case["expected"], "w", encoding="utf-8") as f:
        json.dump(moved, f, ensure_ascii=False)
    rep = R.replay(args)
    ran = [c for c in rep["cases"] if c["status"] != "skipped"]
    assert ran[0]["status"] == "DIFFERENT" and rep["failed"] == 1 and abs(ran[0]["max_abs_dt_word_s"] - 0.3) < 1e-6
    assert os.path.isfile(tmp_path / "dump" / "tiny_auto_bonjour.wav.got.json")


@pytest.mark.gpu
def test_reference_goldens_replay_where_checkpoints_exist():
    """The real thing (BASELINE configs[0]): needs openai-whisper, tiny.en.pt / tiny.pt, ffmpeg, /root/reference's
    tests/ tree (pass WT_REFERENCE_ROOT elsewhere) and an MI355X.  Skips, naming what is missing, anywhere else."""
    R = load_tool()
    args = R.argparse.Namespace(reference=os.environ.get("WT_REFERENCE_ROOT", "/root/reference"),
                                download_root=os.environ.get("WT_WHISPER_CHECKPOINTS"), device="cuda", long=False, dump=None)
    mods_before = sys.modules.get("whisper")
    if mods_before is not None and mods_before.__name__ == "whisper_double":
        for k in [k for k in sys.modules if k == "whisper" or k.startswith("whisper.")]:
            del sys.modules[k]                  # (another test installed the stand-in: look for the real package)
    rep = R.replay(args)
    if rep["replayed"] == 0:
        need = sorted({n for c in rep["cases"] for n in c.get("needs", [])})
        pytest.skip("nothing to replay on this machine; missing: " + ", ".join(need))
    assert rep["failed"] == 0, json.dumps([c for c in rep["cases"] if c["status"] == "DIFFERENT"], indent=1)
