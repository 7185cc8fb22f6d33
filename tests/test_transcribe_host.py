"""Host logic of transcribe() (hook state machine, naive driver, confidence glue,
post-processors) against the REFERENCE'S OWN OUTPUT (tests/golden/transcribe_cases.json,
produced by running /root/reference's transcribe_timestamped on the same whisper double).

No GPU here: the five kernel entry points are replaced by the CPU oracle
(tests/cpu_kernel_standin.py), so what is checked is every decision of the host
layer.  With oracle numerics the result must equal the reference's exactly.
The same cases run with the real HIP kernels in tests/test_gpu_transcribe.py.
"""
import copy
import json
import os

import pytest

import cpu_kernel_standin
from golden import make_golden_transcribe as G

CASES = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "transcribe_cases.json"), encoding="utf-8"))


def run_case(c, device="cpu", raw_confidence=False):
    """raw_confidence: confidences BEFORE the reference's round(, 3) (whisper_timestamped.words.RAW_CONFIDENCE)."""
    import whisper_double as W
    from whisper_double.decoding import Script, set_script
    W.install()
    import whisper_timestamped as wt
    from whisper_timestamped import words
    model, audio, _ = G.build_case(c, device=device)
    script = set_script(Script(c["recorded"]))       # replay exactly what the reference run sampled
    words.RAW_CONFIDENCE = bool(raw_confidence)
    try:
        result = wt.transcribe(model, audio, fp16=False, **c["opts"])
    finally:
        words.RAW_CONFIDENCE = False
        set_script(None)
    assert script.record == c["recorded"]
    return json.loads(json.dumps(G.public_view(result), default=float))


def rounded(view):
    """A raw-confidence view -> what the public call returns (round(, 3) as transcribe.py:1807-1808)."""
    view = copy.deepcopy(view)
    for s in view["segments"]:
        if "confidence" in s:
            s["confidence"] = round(s["confidence"], 3)
        for w in s["words"]:
            if "confidence" in w:
                w["confidence"] = round(w["confidence"], 3)
    return view


def raw_confidence_gap(view, case):
    """max |confidence - reference's confidence| before rounding (BASELINE.json: within 1e-4)."""
    got, want = G.raw_confidences(view), case["expected_raw_confidence"]
    assert len(got) == len(want)
    return max((abs(a - b) for a, b in zip(got, want)), default=0.0)


def raw_logprob_gap(view, case):
    """max |log(confidence) - log(reference's confidence)| = the gap of the per-word / per-segment MEAN LOG-PROBABILITY
    (confidence = exp(mean log-prob), transcribe.py:984-995), read from the raw (unrounded) values.  A random-init model
    without logit filters spreads its mass over the vocabulary (p ~ 1/V: confidences of 1e-8), where |dconfidence| <= 1e-4
    holds whatever the kernel returns; in the log domain those cases are compared for real.  An exact 0.0 (a word with no
    kept token, transcribe.py:989) must be an exact 0.0 on both sides."""
    import math
    got, want = G.raw_confidences(view), case["expected_raw_confidence"]
    assert len(got) == len(want)
    worst = 0.0
    for a, b in zip(got, want):
        assert (a == 0) == (b == 0), (a, b)
        if b:
            assert a > 0 and b > 0, (a, b)
            worst = max(worst, abs(math.log(a) - math.log(b)))
    return worst


LOGPROB_TOL = 2e-4      # mean log-prob of a word / segment, GPU kernels vs the reference's torch CPU log_softmax


def compare(got, exp, time_tol, conf_tol, logprob_tol=1e-4, sampled=False):
    assert got["text"] == exp["text"]
    assert got["language"] == exp["language"]
    assert len(got["segments"]) == len(exp["segments"])
    assert got.get("speech_activity") == exp.get("speech_activity")
    if "language_probs_top" in exp:
        assert list(got["language_probs_top"]) == list(exp["language_probs_top"])
        for k, v in exp["language_probs_top"].items():
            assert abs(got["language_probs_top"][k] - v) <= 1e-4
    worst_t = worst_c = 0.0
    for gs, es in zip(got["segments"], exp["segments"]):
        for k in ("id", "seek", "text", "tokens", "temperature"):
            assert gs.get(k) == es.get(k), (k, gs.get(k), es.get(k))
        # (random sampling with several hypotheses: the backend's avg_logprob belongs to whichever hypothesis its own
        #  RNG produced before the scripted result replaced it -- device dependent, not this repository's output)
        for k in (("no_speech_prob", "compression_ratio") if sampled else ("avg_logprob", "no_speech_prob", "compression_ratio")):
            assert abs(gs[k] - es[k]) <= logprob_tol * max(1.0, abs(es[k])), (k, gs[k], es[k])
        assert ("confidence" in gs) == ("confidence" in es)
        if "confidence" in es:
            worst_c = max(worst_c, abs(gs["confidence"] - es["confidence"]))
        worst_t = max(worst_t, abs(gs["start"] - es["start"]), abs(gs["end"] - es["end"]))
        assert [w["text"] for w in gs["words"]] == [w["text"] for w in es["words"]]
        for gw, ew in zip(gs["words"], es["words"]):
            assert set(gw) == set(ew), (gw, ew)
            worst_t = max(worst_t, abs(gw["start"] - ew["start"]), abs(gw["end"] - ew["end"]))
            if "confidence" in ew:
                worst_c = max(worst_c, abs(gw["confidence"] - ew["confidence"]))
    assert worst_t <= time_tol + 1e-9, f"max |dt| = {worst_t}"
    assert worst_c <= conf_tol + 1e-9, f"max |dconfidence| = {worst_c}"
    return worst_t, worst_c


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_transcribe_host_logic_equals_reference(case, monkeypatch):
    cpu_kernel_standin.install(monkeypatch)
    raw = run_case(copy.deepcopy(case), raw_confidence=True)
    compare(rounded(raw), case["expected"], time_tol=0.0, conf_tol=0.0, logprob_tol=1e-6)
    assert raw_confidence_gap(raw, case) <= 1e-5      # (batched windows: GEMM batch-size rounding, ~2e-6)
    assert raw_logprob_gap(raw, case) <= 2e-5         # the same comparison in the log domain (vacuous nowhere)


def test_every_case_with_confidences_is_compared_where_it_means_something():
    """Each golden either has a confidence > 1e-2 somewhere (the 1e-4 bar bites) or only tiny ones -- in which case the
    log-domain comparison is what holds it.  Both comparisons run on every case; this test documents which cases rely
    on the second (8 of 33, including BASELINE configs[2])."""
    tiny = [c["name"] for c in CASES if c["expected_raw_confidence"] and max(c["expected_raw_confidence"]) < 1e-2]
    assert "small_beam5_table_heads" in tiny and len(tiny) == 8, tiny
    for c in CASES:
        nonzero = [x for x in c["expected_raw_confidence"] if x]
        assert all(x > 1e-30 for x in nonzero)        # log() of every stored value is well inside float range


def is_sampled(case):
    return case["opts"].get("best_of") is not None


EFFICIENT = [c for c in CASES if not (c["opts"].get("naive_approach") or c["opts"].get("beam_size") or c["opts"].get("best_of")
                                      or isinstance(c["opts"].get("temperature"), (list, tuple)))]


@pytest.mark.parametrize("case", EFFICIENT, ids=[c["name"] for c in EFFICIENT])
def test_reusing_the_decoder_logits_gives_the_same_result(case, monkeypatch):
    """Opt-in efficient.REUSE_DECODER_LOGITS: the filtered logits whisper's sampler leaves behind instead of a second
    projection + filter pass per token.  Same words and times; confidences equal up to GEMM-vs-GEMV rounding."""
    from whisper_timestamped import efficient
    cpu_kernel_standin.install(monkeypatch)
    monkeypatch.setattr(efficient, "REUSE_DECODER_LOGITS", True)
    got = run_case(copy.deepcopy(case))
    compare(got, case["expected"], time_tol=0.0, conf_tol=1e-3 + 1e-9, logprob_tol=1e-5)


def test_logits_reuse_verifies_itself_window_by_window(monkeypatch):
    """efficient.REUSE_DECODER_LOGITS = "auto" (default): every 30 s window's first token is done both ways; with a backend
    that filters in place the rest of the window reuses the decoder's rows, and one alignment launch set per window."""
    from whisper_timestamped import efficient
    cpu_kernel_standin.install(monkeypatch)
    case = next(c for c in CASES if c["name"] == "three_windows_single_segment_window")
    got = run_case(copy.deepcopy(case))
    compare(got, case["expected"], time_tol=0.0, conf_tol=0.0, logprob_tol=1e-6)
    assert efficient.LAST_SESSION["reuse_mode"] == "auto" and not efficient.LAST_SESSION["reuse_fell_back"]
    assert efficient.LAST_SESSION["windows_verified"] == 3 and efficient.LAST_SESSION["alignment_launch_sets"] == 3


def test_logits_reuse_falls_back_when_the_backend_filters_a_copy(monkeypatch):
    """A backend whose sampler filters a COPY of the decoder's logits: the first token's comparison fails, the session
    goes back to the reference's per-token projection + filters -- and the result is still the reference's."""
    import whisper_double as W
    from whisper_timestamped import efficient
    cpu_kernel_standin.install(monkeypatch)
    W.install()
    inner = W.decoding.PyTorchInference.logits
    monkeypatch.setattr(W.decoding.PyTorchInference, "logits", lambda self, tokens, af: inner(self, tokens, af).clone())
    for name in ("two_windows_prompted", "decoding_limit"):
        case = next(c for c in CASES if c["name"] == name)
        got = run_case(copy.deepcopy(case))
        compare(got, case["expected"], time_tol=0.0, conf_tol=0.0, logprob_tol=1e-6)
        assert efficient.LAST_SESSION["reuse_fell_back"] and efficient.LAST_SESSION["reuse_state"] == "off"
    monkeypatch.setattr(efficient, "REUSE_DECODER_LOGITS", True)          # forced reuse on such a backend: refused loudly
    with pytest.raises(RuntimeError, match="does not filter the decoder's logits in place"):
        run_case(copy.deepcopy(next(c for c in CASES if c["name"] == "two_windows_prompted")))


def test_the_reference_way_of_computing_the_logits(monkeypatch):
    """efficient.REUSE_DECODER_LOGITS = False: a second projection + filter pass per token, exactly as the reference."""
    from whisper_timestamped import efficient
    cpu_kernel_standin.install(monkeypatch)
    monkeypatch.setattr(efficient, "REUSE_DECODER_LOGITS", False)
    for name in ("two_windows_prompted", "no_speech_skip", "language_detection", "eot_without_end_timestamp"):
        case = next(c for c in CASES if c["name"] == name)
        got = run_case(copy.deepcopy(case))
        compare(got, case["expected"], time_tol=0.0, conf_tol=0.0, logprob_tol=1e-6)
        assert efficient.LAST_SESSION["windows_verified"] == 0


def test_segment_by_segment_alignment_gives_the_same_result(monkeypatch):
    """efficient.DEFER_ALIGNMENT = False: one synchronous alignment per flushed segment (the reference's shape) instead of
    one launch set per window read a window later -- same output."""
    from whisper_timestamped import efficient
    cpu_kernel_standin.install(monkeypatch)
    monkeypatch.setattr(efficient, "DEFER_ALIGNMENT", False)
    for name in ("two_windows_prompted", "no_trust_whisper_timestamps", "decoding_limit", "no_speech_skip"):
        case = next(c for c in CASES if c["name"] == name)
        got = run_case(copy.deepcopy(case))
        compare(got, case["expected"], time_tol=0.0, conf_tol=0.0, logprob_tol=1e-6)


def test_naive_strategy_prints_words_when_verbose(monkeypatch, capsys):
    """The reference prints every word inside the naive loop when verbose is truthy (transcribe.py:1304-1305) -- with the
    window-by-window second pass and with the batched one."""
    import re
    cpu_kernel_standin.install(monkeypatch)
    for name in ("naive_greedy", "naive_no_trust_three_windows"):
        case = copy.deepcopy(next(c for c in CASES if c["name"] == name))
        case["opts"]["verbose"] = True
        got = run_case(case)
        shown = [l for l in capsys.readouterr().out.splitlines() if re.match(r"^\[\d\d:\d\d\.\d{3} --> \d\d:\d\d\.\d{3}\] ", l)]
        n_words = sum(len(s["words"]) for s in got["segments"])
        assert n_words > 0 and len([l for l in shown if not l.endswith("] ")]) >= n_words
