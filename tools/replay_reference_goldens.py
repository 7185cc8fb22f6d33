#!/usr/bin/env python3
"""Replay the REFERENCE'S OWN end-to-end goldens through this repository (BASELINE configs[0]).

The reference's test-suite (/root/reference/tests/test_transcribe.py) runs its command line / Python API with real
Whisper checkpoints on the clips under tests/data and compares the produced ``*.words.json`` with the files under
tests/expected/ -- loosely: every float rounded to ONE decimal (`loose`, test_transcribe.py:185-198).  This script
does the same with THIS repository's ``whisper_timestamped`` (alignment on the MI355X), for the cases that pin the hot
path:

  tiny.en          cli --model tiny.en {--efficient | --accurate | --condition False --efficient}
                   on bonjour_vous_allez_bien.mp3                       (test_monolingual_tiny, :454-477)
  tiny_auto        transcribe(load_model("tiny"), f) for bonjour.wav, laugh1.mp3, laugh2.mp3   (test_python_import, :704-711)
  tiny_fr          the same with language="fr"                                                   (:713-717)
  tiny_auto/batch  the three tiny_auto recordings through ONE transcribe_batch call (B decoder streams: not a reference
  tiny_fr/batch    test, the same goldens), and the three tiny_fr ones
  naive            cli --model small --language en {--naive | --accurate} on apollo11.mp3 (long; only with --long)

It needs what this image does not have: the openai-whisper package, the checkpoints (tiny.en.pt, tiny.pt, [small.pt]
under --download_root or ~/.cache/whisper) and, for the mp3 clips, ffmpeg.  Whatever is missing is listed and the
case is SKIPPED (exit status 0 with "replayed: 0" -- nothing can be verified here); a case that runs and differs is a
FAILURE (exit status 1).  Besides the reference's own loose comparison the report gives what BASELINE.json asks for:
max |dt| over all word boundaries and max |dconfidence| over all words (bars 0.02 s / 1e-3 on the rounded values the
goldens hold), when the two outputs have the same words.

    python tools/replay_reference_goldens.py --reference /root/reference [--download_root DIR] [--device cuda] [--long]
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "whisper-timestamped_amd"))


def loose(obj):
    """The reference's tolerance (tests/test_transcribe.py:185-198): floats to one decimal."""
    if isinstance(obj, list):
        return [loose(a) for a in obj]
    if isinstance(obj, float):
        f = round(obj, 1)
        return 0.0 if f == -0.0 else f
    if isinstance(obj, dict):
        return {k: loose(v) for k, v in obj.items()}
    if isinstance(obj, tuple):
        return tuple(loose(list(obj)))
    return obj


def norm_language(content):
    if isinstance(content, dict) and "language" in content:
        content["language"] = {"japanese": "ja"}.get(str(content["language"]).lower(), content["language"])
    return content


def word_gaps(got, want):
    """(max |dt|, max |dconfidence|, comparable?) over the words of two result dictionaries."""
    gw = [w for s in got.get("segments", []) for w in s.get("words", [])]
    ww = [w for s in want.get("segments", []) for w in s.get("words", [])]
    if [w["text"] for w in gw] != [w["text"] for w in ww]:
        return None, None, False
    dt = max((max(abs(a["start"] - b["start"]), abs(a["end"] - b["end"])) for a, b in zip(gw, ww)), default=0.0)
    dc = max((abs(a["confidence"] - b["confidence"]) for a, b in zip(gw, ww) if "confidence" in a and "confidence" in b),
             default=0.0)
    return dt, dc, True


def requirements(args):
    """What is present / missing on this machine, checked once."""
    have = {}
    try:
        import whisper  # noqa: F401
        # (tests/whisper_double registers itself as `whisper` inside the test-suite: a random-weight stand-in is not
        #  the package the goldens were produced with)
        have["openai-whisper"] = None if whisper.__name__ == "whisper_double" else getattr(whisper, "__version__", "?")
    except ImportError:
        have["openai-whisper"] = None
    have["ffmpeg"] = shutil.which("ffmpeg")
    root = args.download_root or os.path.join(os.path.expanduser("~"), ".cache", "whisper")
    for name in ("tiny.en", "tiny", "small"):
        p = os.path.join(root, name + ".pt")
        have[f"checkpoint {name}"] = p if os.path.isfile(p) else None
    try:
        import torch
        have["gpu"] = torch.cuda.get_device_name(0) if torch.cuda.is_available() else None
    except Exception:                                             # noqa: BLE001
        have["gpu"] = None
    lib = os.path.join(ROOT, "whisper-timestamped_amd", "libwtalign.so")
    have["libwtalign.so"] = lib if os.path.isfile(lib) else None
    return have


def cases(args):
    data = os.path.join(args.reference, "tests", "data")
    exp = os.path.join(args.reference, "tests", "expected")
    out = []
    clip = os.path.join(data, "bonjour_vous_allez_bien.mp3")
    for prefix, opts in (("efficient", ["--efficient"]), ("accurate", ["--accurate"]),
                         ("nocond", ["--condition", "False", "--efficient"])):
        out.append(dict(name=f"tiny.en/{prefix}", kind="cli", model="tiny.en", audio=clip, opts=["--model", "tiny.en", *opts],
                        expected=os.path.join(exp, "tiny.en", f"{prefix}_bonjour_vous_allez_bien.mp3.words.json")))
    for sub, kw in (("tiny_auto", {}), ("tiny_fr", {"language": "fr"})):
        for fn in ("bonjour.wav", "laugh1.mp3", "laugh2.mp3"):
            out.append(dict(name=f"{sub}/{fn}", kind="api", model="tiny", audio=os.path.join(data, fn), kwargs=kw,
                            expected=os.path.join(exp, sub, f"{fn}.words.json")))
    for sub, kw in (("tiny_auto", {}), ("tiny_fr", {"language": "fr"})):
        fns = ("bonjour.wav", "laugh1.mp3", "laugh2.mp3")
        out.append(dict(name=f"{sub}/batch", kind="api_batch", model="tiny", audio=[os.path.join(data, fn) for fn in fns], kwargs=kw,
                        expected=[os.path.join(exp, sub, f"{fn}.words.json") for fn in fns]))
    if args.long:
        clip = os.path.join(data, "apollo11.mp3")
        for prefix, opts in (("naive", ["--naive"]), ("accurate", ["--accurate"])):   # (reference: test_naive, :332-346)
            out.append(dict(name=f"naive/{prefix}", kind="cli", model="small", audio=clip,
                            opts=["--model", "small", "--language", "en", *opts],
                            expected=os.path.join(exp, "naive", f"{prefix}_apollo11.mp3.words.json")))
    return out


def missing_for(case, have):
    miss = [k for k in ("openai-whisper", "gpu", "libwtalign.so", f"checkpoint {case['model']}") if not have.get(k)]
    audios = case["audio"] if isinstance(case["audio"], list) else [case["audio"]]
    expected = case["expected"] if isinstance(case["expected"], list) else [case["expected"]]
    if any(not a.endswith(".wav") for a in audios) and not have["ffmpeg"]:
        miss.append("ffmpeg")
    for path in audios + expected:
        if not os.path.isfile(path):
            miss.append(path)
    return miss


def run_case(case, args, models):
    import whisper_timestamped as wt
    if case["kind"] == "api_batch":
        if case["model"] not in models:
            models[case["model"]] = wt.load_model(case["model"], device=args.device, download_root=args.download_root)
        res = wt.transcribe_batch(models[case["model"]], case["audio"], **case["kwargs"])
        return [json.loads(json.dumps(r, ensure_ascii=False, default=float)) for r in res]
    if case["kind"] == "api":
        if case["model"] not in models:
            models[case["model"]] = wt.load_model(case["model"], device=args.device, download_root=args.download_root)
        res = wt.transcribe(models[case["model"]], case["audio"], **case["kwargs"])
        return json.loads(json.dumps(res, ensure_ascii=False, default=float))
    from whisper_timestamped.cli import cli
    out_dir = tempfile.mkdtemp(prefix="wt_replay_")
    try:
        extra = ["--model_dir", args.download_root] if args.download_root else []
        cli([case["audio"], "--output_dir", out_dir, "--device", args.device, *case["opts"], *extra])
        with open(os.path.join(out_dir, os.path.basename(case["audio"]) + ".words.json"), encoding="utf-8") as f:
            return json.load(f)
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)


def replay(args):
    have = requirements(args)
    report = {"requirements": have, "cases": [], "replayed": 0, "failed": 0, "skipped": 0}
    models = {}
    for case in cases(args):
        miss = missing_for(case, have)
        rec = {"case": case["name"]}
        if miss:
            rec.update(status="skipped", needs=miss)
            report["skipped"] += 1
        else:
            got = run_case(case, args, models)
            if case["kind"] == "api_batch":              # several recordings, one call: every one against its own golden
                wants = [norm_language(json.load(open(e, encoding="utf-8"))) for e in case["expected"]]
                gots = [norm_language(g) for g in got]
                same = all(loose(g) == loose(w_) for g, w_ in zip(gots, wants)) and len(gots) == len(wants)
                gaps = [word_gaps(g, w_) for g, w_ in zip(gots, wants)]
                comparable = all(g[2] for g in gaps)
                dt = max((g[0] for g in gaps), default=0.0) if comparable else None
                dc = max((g[1] for g in gaps), default=0.0) if comparable else None
                got, want = {"recordings": gots}, {"recordings": wants}
            else:
                got = norm_language(got)
                want = norm_language(json.load(open(case["expected"], encoding="utf-8")))
                same = loose(got) == loose(want)
                dt, dc, comparable = word_gaps(got, want)
            rec.update(status="ok" if same else "DIFFERENT", reference_loose_comparison=same, same_words=comparable,
                       max_abs_dt_word_s=dt, max_abs_dconfidence=dc,
                       within_baseline_bars=bool(comparable and dt <= 0.02 + 1e-9 and dc <= 1e-3 + 1e-9))
            report["replayed"] += 1
            report["failed"] += 0 if same else 1
            if not same and args.dump:
                os.makedirs(args.dump, exist_ok=True)
                with open(os.path.join(args.dump, case["name"].replace("/", "_") + ".got.json"), "w", encoding="utf-8") as f:
                    json.dump(got, f, indent=2, ensure_ascii=False)
        report["cases"].append(rec)
    return report


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--reference", default="/root/reference", help="checkout of linto-ai/whisper-timestamped (tests/data, tests/expected)")
    ap.add_argument("--download_root", default=None, help="directory holding tiny.en.pt / tiny.pt / small.pt (default ~/.cache/whisper)")
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--long", action="store_true", help="also the apollo11.mp3 cases (whisper-small, minutes)")
    ap.add_argument("--dump", default=None, help="directory for the outputs of the cases that differ")
    args = ap.parse_args(argv)
    report = replay(args)
    print(json.dumps(report, indent=2, ensure_ascii=False))
    if report["replayed"] == 0:
        need = sorted({n for c in report["cases"] for n in c.get("needs", [])})
        print("replayed: 0 -- nothing could be verified on this machine; missing: " + ", ".join(need), file=sys.stderr)
    return 1 if report["failed"] else 0


if __name__ == "__main__":
    sys.exit(main())
