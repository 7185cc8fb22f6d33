"""The command line (whisper_timestamped/cli.py): the reference's options and output files
(/root/reference/whisper_timestamped/transcribe.py:2964-3182).  No trained checkpoint exists offline, so load_model is
pointed at a random-weight tiny model; kernels = the CPU oracle (tests/cpu_kernel_standin.py)."""
import json
import os

import numpy as np
import pytest

import schema_check

HERE = os.path.dirname(os.path.abspath(__file__))


def _wav(path, seconds=2.5, seed=0):
    from scipy.io import wavfile
    rng = np.random.RandomState(seed)
    t = np.arange(int(seconds * 16000)) / 16000.0
    x = 0.05 * rng.standard_normal(t.shape) + 0.1 * np.sin(2 * np.pi * 220.0 * t)
    wavfile.write(path, 16000, (x * 32767).astype(np.int16))


def _patch(monkeypatch):
    import cpu_kernel_standin
    import whisper_double as W
    W.install()
    cpu_kernel_standin.install(monkeypatch)
    from whisper_timestamped import cli as C
    monkeypatch.setattr(C, "load_model", lambda name, device=None, download_root=None, backend=None:
                        W.build_model("tiny" if name in ("tiny", "small") else name, seed=0, device="cpu"))
    return C


def test_cli_writes_every_output_format(tmp_path, monkeypatch):
    C = _patch(monkeypatch)
    wav = tmp_path / "clip.wav"
    _wav(str(wav))
    out = tmp_path / "out"
    C.cli([str(wav), "--model", "tiny", "--device", "cpu", "--language", "en", "--output_dir", str(out), "--fp16", "False",
           "--punctuations_with_words", "False", "--accurate", "--efficient"])
    names = sorted(os.listdir(out))
    assert names == sorted("clip.wav" + s for s in (".words.json", ".txt", ".vtt", ".words.vtt", ".srt", ".words.srt", ".csv",
                                                    ".words.csv", ".tsv", ".words.tsv"))
    result = json.load(open(out / "clip.wav.words.json", encoding="utf-8"))
    schema_check.validate(result, json.load(open(os.path.join(HERE, "golden", "json_schema.json"))))
    n_words = sum(len(s["words"]) for s in result["segments"])
    assert (out / "clip.wav.vtt").read_text().startswith("WEBVTT\n")
    assert (out / "clip.wav.words.srt").read_text().count(" --> ") == n_words
    assert len((out / "clip.wav.words.csv").read_text().strip().splitlines()) == n_words
    assert (out / "clip.wav.tsv").read_text().splitlines()[0].split("\t") == ["start", "end", "text"]


def test_cli_prints_filtered_json_without_output_dir(tmp_path, monkeypatch, capsys):
    C = _patch(monkeypatch)
    wav = tmp_path / "clip.wav"
    _wav(str(wav), seconds=1.5, seed=1)
    C.cli([str(wav), "--model", "tiny", "--device", "cpu", "--language", "en", "--fp16", "False", "--output_format", "json,srt",
           "--naive", "--recompute_all_timestamps", "True"])
    shown = json.loads(capsys.readouterr().out)
    assert set(shown) <= {"text", "segments", "language", "language_probs", "speech_activity"}
    for seg in shown["segments"]:
        assert set(seg) <= {"text", "start", "end", "confidence", "words"}


def test_cli_options_are_the_reference_ones():
    """Option names and defaults of the reference's parser (transcribe.py:3000-3079)."""
    import whisper_double as W
    W.install()
    from whisper_timestamped import cli as C
    p = C.build_parser()
    d = {a.dest: a.default for a in p._actions}
    want = dict(model="small", model_dir=None, backend="openai-whisper", output_dir=None, output_format="all", task="transcribe",
                language=None, vad=False, detect_disfluencies=False, recompute_all_timestamps=False, punctuations_with_words=True,
                temperature=0.0, best_of=None, beam_size=None, patience=None, length_penalty=None, suppress_tokens="-1",
                initial_prompt=None, condition_on_previous_text=True, fp16=None, temperature_increment_on_fallback=0.0,
                compression_ratio_threshold=2.4, logprob_threshold=-1.0, no_speech_threshold=0.6, threads=0,
                compute_confidence=True, verbose=False, plot=False, debug=False, naive=False)
    for k, v in want.items():
        assert d[k] == v, (k, d[k], v)
    with pytest.raises(ValueError):
        C._output_formats("json,doc")


def test_cli_streams_writes_the_same_files_as_one_after_the_other(tmp_path, monkeypatch):
    """--streams N (not in the reference): several audio files through ONE transcribe_batch call; the output files must be
    those of the default, one-file-after-the-other run."""
    from test_streams_host import install_streams_standin
    C = _patch(monkeypatch)
    install_streams_standin(monkeypatch)
    wavs = []
    for k, seconds in enumerate((2.5, 4.0, 1.2)):
        w = tmp_path / f"clip{k}.wav"
        _wav(str(w), seconds=seconds, seed=k)
        wavs.append(str(w))
    common = ["--model", "tiny", "--device", "cpu", "--language", "en", "--fp16", "False", "--output_format", "json,srt", "--efficient"]
    C.cli([*wavs, "--output_dir", str(tmp_path / "serial"), *common])
    C.cli([*wavs, "--output_dir", str(tmp_path / "streams"), "--streams", "2", *common])
    names = sorted(os.listdir(tmp_path / "serial"))
    assert names == sorted(os.listdir(tmp_path / "streams")) and len(names) == 9
    for n in names:
        a, b = (tmp_path / "serial" / n).read_text(), (tmp_path / "streams" / n).read_text()
        if n.endswith(".json"):
            ja, jb = json.loads(a), json.loads(b)
            assert ja["text"] == jb["text"] and len(ja["segments"]) == len(jb["segments"])
            for sa, sb in zip(ja["segments"], jb["segments"]):
                assert [(w["text"], w["start"], w["end"]) for w in sa["words"]] == [(w["text"], w["start"], w["end"]) for w in sb["words"]]
                assert all(abs(x["confidence"] - y["confidence"]) <= 1e-3 + 1e-9 for x, y in zip(sa["words"], sb["words"]))
        else:
            assert a == b, n


def test_cli_streams_verbose_without_output_dir_still_prints_the_results(tmp_path, monkeypatch, capsys):
    """--streams N --verbose True without --output_dir: the B-stream path prints no segments while it decodes,
    so its results are dumped as JSON whatever --verbose says -- one document per recording, as in the non-verbose form."""
    from test_streams_host import install_streams_standin
    C = _patch(monkeypatch)
    install_streams_standin(monkeypatch)
    wavs = []
    for k, seconds in enumerate((2.0, 3.0)):
        w = tmp_path / f"clip{k}.wav"
        _wav(str(w), seconds=seconds, seed=10 + k)
        wavs.append(str(w))
    common = ["--model", "tiny", "--device", "cpu", "--language", "en", "--fp16", "False", "--efficient"]
    C.cli([*wavs, "--streams", "2", "--verbose", "True", *common])
    verbose_out = capsys.readouterr().out
    C.cli([*wavs, "--streams", "2", *common])
    quiet_out = capsys.readouterr().out
    assert verbose_out.count('"segments"') == 2 and quiet_out.count('"segments"') == 2
    dec = json.JSONDecoder()
    docs, pos = [], verbose_out.index("{")
    while pos < len(verbose_out):
        obj, end = dec.raw_decode(verbose_out, pos)
        docs.append(obj)
        nxt = verbose_out.find("{", end)
        if nxt < 0:
            break
        pos = nxt
    assert len(docs) == 2 and all("segments" in d_ and "text" in d_ for d_ in docs)
