"""transcribe()'s result dictionary against the REFERENCE'S OWN JSON schema (tests/json_schema.json, what its
tests/test_transcribe.py:287-296 checks every output with), on the reference's own wav fixtures.

BASELINE.json configs[0] (whisper-tiny.en, single clip, greedy, tests/data) can only be run offline with random
weights: there are no trained checkpoints in this image, so the VALUES of the goldens under tests/expected cannot be
replayed -- what can be pinned is the plumbing (a wav PATH goes in, the reference-shaped dictionary comes out) and the
schema.  CPU leg: kernels replaced by the oracle (tests/cpu_kernel_standin.py), reads /root/reference directly.
GPU leg (-m gpu): real kernels, synthetic audio, the committed copy of the schema (tests/golden/json_schema.json).
"""
import json
import os

import numpy as np
import pytest
import torch

import schema_check

HERE = os.path.dirname(os.path.abspath(__file__))
REF_TESTS = "/root/reference/tests"


def _models():
    import whisper_double as W
    W.install()
    return W


def _check(result, schema):
    plain = json.loads(json.dumps(result, default=float))          # what json.dump writes (numpy floats -> numbers)
    schema_check.validate(plain, schema)
    for seg in plain["segments"]:                                  # and what the schema leaves open
        assert {"id", "seek", "start", "end", "text", "tokens", "temperature", "avg_logprob", "compression_ratio",
                "no_speech_prob"} <= set(seg)
        for w in seg.get("words", []):
            assert {"text", "start", "end"} <= set(w) and w["end"] >= w["start"]
    return plain


def test_committed_schema_is_the_reference_one():
    if not os.path.isfile(os.path.join(REF_TESTS, "json_schema.json")):
        pytest.skip("/root/reference is not present")
    assert json.load(open(os.path.join(HERE, "golden", "json_schema.json"))) == \
        json.load(open(os.path.join(REF_TESTS, "json_schema.json")))


def test_validator_rejects_what_the_schema_forbids():
    schema = json.load(open(os.path.join(HERE, "golden", "json_schema.json")))
    ok = {"text": "a", "language": "en", "segments": [{"id": 0, "start": 0.0, "end": 1.0, "text": "a", "tokens": [1, 2],
                                                        "temperature": 0.0, "avg_logprob": -0.5, "compression_ratio": 1.0,
                                                        "no_speech_prob": 0.1, "confidence": 0.5,
                                                        "words": [{"text": "a", "start": 0.0, "end": 1.0, "confidence": 0.5}]}]}
    schema_check.validate(ok, schema)
    for path, bad in ((("segments", 0, "confidence"), 1.5), (("segments", 0, "tokens", 0), 60000), (("segments", 0, "start"), -1.0),
                      (("segments", 0, "words", 0, "text"), 3), (("text",), None), (("segments", 0, "avg_logprob"), 0.2)):
        doc = json.loads(json.dumps(ok))
        node = doc
        for k in path[:-1]:
            node = node[k]
        node[path[-1]] = bad
        with pytest.raises(schema_check.SchemaError):
            schema_check.validate(doc, schema)


@pytest.mark.parametrize("wav", ["bonjour.wav", "empty.wav"])
@pytest.mark.parametrize("model_name,opts", [("tiny.en", dict(language="en")), ("tiny", dict()),
                                             ("tiny", dict(language="fr", naive_approach=True, trust_whisper_timestamps=False))],
                         ids=["tiny.en", "tiny-detect-language", "tiny-naive-batched"])
def test_reference_wav_fixtures_validate_against_reference_schema(wav, model_name, opts, monkeypatch):
    """configs[0] plumbing: wav path -> load_audio -> greedy decode of a random-weight tiny(.en) -> hooks -> alignment ->
    the reference-shaped dictionary, valid under the reference's schema."""
    path = os.path.join(REF_TESTS, "data", wav)
    if not os.path.isfile(path):
        pytest.skip("/root/reference is not present")
    import cpu_kernel_standin
    cpu_kernel_standin.install(monkeypatch)
    W = _models()
    import whisper_timestamped as wt
    model = W.build_model(model_name, seed=0, device="cpu")
    result = wt.transcribe(model, path, fp16=False, sample_len=24, **opts)      # (sample_len: a random model never says <|eot|>)
    schema = json.load(open(os.path.join(REF_TESTS, "json_schema.json")))
    plain = _check(result, schema)
    assert plain["language"] == opts.get("language", plain["language"])
    if "language" not in opts:
        assert abs(sum(plain["language_probs"].values()) - 1.0) < 1e-3
    duration = W.load_audio(path).shape[0] / 16000
    for seg in plain["segments"]:
        for w in seg.get("words", []):
            assert w["end"] <= max(duration, 30.0) + 0.02


@pytest.mark.gpu
def test_gpu_results_validate_against_reference_schema():
    """The same schema on the MI355X (real kernels): an unscripted greedy run, the batched naive path, and language
    detection, on synthetic audio."""
    W = _models()
    import whisper_timestamped as wt
    schema = json.load(open(os.path.join(HERE, "golden", "json_schema.json")))
    g = torch.Generator().manual_seed(5)
    t = torch.arange(int(41.5 * 16000)) / 16000.0
    audio = (0.05 * torch.randn(t.shape, generator=g) + 0.1 * torch.sin(2 * np.pi * 220.0 * t)).float()
    for name, opts in (("tiny.en", dict(language="en")), ("tiny", dict()),
                       ("tiny", dict(language="en", naive_approach=True, trust_whisper_timestamps=False)),
                       ("tiny", dict(language="en", beam_size=2, detect_disfluencies=True))):
        model = W.build_model(name, seed=0, device="cuda:0")
        result = wt.transcribe(model, audio, fp16=False, sample_len=24, **opts)
        plain = _check(result, schema)
        assert len(plain["segments"]) > 0


def test_pipeline_schedule_rule_and_lanes():
    """whisper_timestamped.pipeline: every stage of the step has a lane; `auto` = hilo up to 128 units with separate cost / DTW
    entries, serial beyond or with the fused small-unit entry; an explicit choice is kept; unknown names are refused."""
    import pytest
    from whisper_timestamped import pipeline as P
    assert set(P.STAGES) <= set(P.LANE) and {P.LANE[s] for s in P.STAGES} == {"hi", "lo"}
    assert P.LANE["dtw"] == P.LANE["logmel"] == "hi" and P.LANE["cost"] == P.LANE["logprob"] == "lo"
    assert P.choose_schedule("auto", 32) == "hilo" and P.choose_schedule("auto", 128) == "hilo"
    assert P.choose_schedule("auto", 256) == "serial" and P.choose_schedule("auto", 32, fused_small_units=True) == "serial"
    assert P.choose_schedule("serial", 32) == "serial" and P.choose_schedule("hilo", 256) == "hilo"
    with pytest.raises(AssertionError):
        P.choose_schedule("two_streams", 32)


def test_bench_word_gap_bookkeeping():
    """bench.word_gaps / merge_gaps / gaps_ok_between_batch_sizes: the per-word parity rule of the transcribe()-level legs
    (texts equal, confidences within 1e-4, mean log-probabilities within 2e-4, times within 0.02 s for >= 99 % of the words)."""
    import math
    import wordgaps as bench
    a = [("w%d" % i, 0.5 * i, 0.5 * i + 0.4, 0.01 * (1 + i % 7)) for i in range(300)]
    same = bench.word_gaps(a, list(a), "same")
    assert same == [0.0, 0.0, 0.0, 300, 0] and bench.gaps_ok_between_batch_sizes(same)
    b = list(a)
    b[10] = (a[10][0], a[10][1] + 0.22, a[10][2], a[10][3] * math.exp(1e-5))
    g = bench.word_gaps(a, b, "one word moved")
    assert abs(g[0] - 0.22) < 1e-12 and g[3:] == [300, 1] and 0 < g[2] < 2e-5
    assert bench.gaps_ok_between_batch_sizes(g)                                   # 1 of 300 <= 1 %
    for k in (20, 30, 40, 50):
        b[k] = (a[k][0], a[k][1], a[k][2] + 0.04, a[k][3])
    g = bench.word_gaps(a, b, "five words moved")
    assert g[4] == 5 and not bench.gaps_ok_between_batch_sizes(g)                 # 5 of 300 > 1 %
    c = list(a)
    c[3] = (a[3][0], a[3][1], a[3][2], a[3][3] + 2e-4)
    assert not bench.gaps_ok_between_batch_sizes(bench.word_gaps(a, c, "confidence off"))
    m = bench.merge_gaps(same, g)
    assert m[3] == 600 and m[4] == 5 and bench.gaps_report(m)["words_beyond_0.02_s"] == 5
    import pytest
    with pytest.raises(AssertionError):
        bench.word_gaps(a, [("x",) + t[1:] for t in a], "texts differ")
