"""transcribe_batch (whisper_timestamped/streams.py) on the MI355X: B recordings per decoder op, real kernels
(wt_qk_rows_streams, wt_align_batch_v3, wt_logprob_gather_rows, wt_find_start_padding_batch, wt_logmel_batch).

Bars (BASELINE.json north_star, as tests/test_gpu_transcribe.py): against the REFERENCE's output for every recording --
word times within 0.02 s, confidences within 1e-4 before rounding, mean log-probabilities within 2e-4 -- and against
this repository's own one-stream path: a batch of eight equals eight single calls word for word and time for time.
"""
import copy
import json

import numpy as np

import pytest
import torch

from golden import make_golden_transcribe as G
from test_streams_host import BATCHABLE, run_batch, same_model_cases
from test_transcribe_host import CASES, LOGPROB_TOL, compare, is_sampled, raw_confidence_gap, raw_logprob_gap, rounded, run_case

pytestmark = pytest.mark.gpu


def _check_against_reference(raw, case):
    compare(rounded(raw), case["expected"], time_tol=0.02, conf_tol=1e-3 + 1e-4, logprob_tol=2e-4, sampled=is_sampled(case))
    assert raw_confidence_gap(raw, case) <= 1e-4
    assert raw_logprob_gap(raw, case) <= LOGPROB_TOL


@pytest.mark.parametrize("case", BATCHABLE, ids=[c["name"] for c in BATCHABLE])
def test_a_batch_of_one_stream_matches_the_reference(case):
    _check_against_reference(run_batch([copy.deepcopy(case)], device="cuda:0")[0], case)


def test_eight_ragged_streams_each_match_the_reference():
    from whisper_timestamped import streams
    cases = same_model_cases() + same_model_cases()[::-1]
    for raw, case in zip(run_batch(cases, device="cuda:0"), cases):
        _check_against_reference(raw, case)
    assert streams.LAST_RUN["streams"] == 8 and streams.LAST_RUN["alignment_launch_sets"] <= streams.LAST_RUN["rounds"] + 1


def test_a_batch_of_eight_equals_eight_single_calls():
    """The same recordings through transcribe() one at a time (B = 1: the backend's own loop, live hooks) and through
    transcribe_batch: same texts, same segmentation, same word times (exactly), confidences within GEMM batch-size
    rounding."""
    cases = same_model_cases() + same_model_cases()
    singles = [run_case(copy.deepcopy(c), device="cuda:0", raw_confidence=True) for c in cases]
    batch = run_batch(cases, device="cuda:0")
    for b, s in zip(batch, singles):
        compare(b, s, time_tol=0.0, conf_tol=2e-5, logprob_tol=1e-4)


@pytest.mark.parametrize("seed,extra_opts", [(2024, None), (31, {"trust_whisper_timestamps": False}), (32, {"condition_on_previous_text": False})],
                         ids=["defaults", "no_trust", "no_condition"])
def test_random_scripts_streams_equal_one_stream_at_a_time_on_the_gpu(monkeypatch, seed, extra_opts):
    """The random-script comparison of tests/test_streams_host.py with the real kernels: eight recordings of one to three
    windows through three ring blocks (continuous admission: blocks are handed on while other streams go on) against
    transcribe() one at a time -- same words, same times."""
    from test_streams_host import test_random_scripts_streams_equal_one_stream_at_a_time as run
    run(monkeypatch, seed, extra_opts, device="cuda:0", time_tol=0.0)


def test_half_precision_model_streams_stay_close():
    """fp16=True (the reference's GPU default): the batch against its own one-stream run, same precision."""
    import whisper_double as W
    from whisper_double.decoding import Script, set_row_scripts, set_script
    W.install()
    import whisper_timestamped as wt
    from whisper_timestamped import streams
    cases = same_model_cases()[:3]
    model, _, _ = G.build_case(cases[0], device="cuda:0")
    audios = [G.build_case(c, device="cuda:0")[1] for c in cases]
    singles = []
    for c, a in zip(cases, audios):
        set_script(Script(c["recorded"]))
        try:
            singles.append(json.loads(json.dumps(G.public_view(wt.transcribe(model, a, fp16=True, **c["opts"])), default=float)))
        finally:
            set_script(None)
    scripts = [Script(c["recorded"]) for c in cases]

    def on_group(idx):
        for i in idx:
            scripts[i].begin_window()
        set_row_scripts([scripts[i] for i in idx])
    streams.ON_GROUP_DECODE = on_group
    try:
        batch = wt.transcribe_batch(model, audios, fp16=True, **cases[0]["opts"])
    finally:
        streams.ON_GROUP_DECODE = None
        set_row_scripts(None)
    for b, s in zip(batch, singles):
        b = json.loads(json.dumps(G.public_view(b), default=float))
        assert [w["text"] for x in b["segments"] for w in x["words"]] == [w["text"] for x in s["segments"] for w in x["words"]]
        dts = [abs(a[k] - c[k]) for x, y in zip(b["segments"], s["segments"]) for a, c in zip(x["words"], y["words"])
               for k in ("start", "end")]
        # Half precision is not reproducible across GEMM shapes (a GEMV for one stream, a GEMM for three: other
        # accumulation orders, 2^-11 relative steps in q and K), and a random-init model's attention is nearly flat, so
        # a path can move where two columns tie within that noise: most boundaries identical, none of them far away.
        close = sum(d <= 0.04 + 1e-9 for d in dts)
        assert close >= 0.85 * len(dts) and max(dts) <= 1.0, (close, len(dts), max(dts))


@pytest.mark.gpu
def test_logprob_digest_streams_kernel_against_the_full_rows():
    """wt_logprob_digest_streams (the real kernel through the C ABI) under the same checks as the host test's stand-in,
    and: record [0] is BIT-identical to wt_logprob_gather_batch on the same rows and tokens -- rows at the same
    address alignment, as check_logits_view lays them out (the last query row of a (g, 3, V) block): both kernels start
    their vector stream on the row's first 128-byte boundary, so the alignment decides which thread sums which logits."""
    from test_streams_host import check_logits_view
    from whisper_timestamped import _lib
    rings, full, sampled, host = check_logits_view("cuda:0")
    n_steps, g, V = full.shape
    for step in range(n_steps):
        outs = torch.zeros((g, 3, V))
        outs[:, -1] = full[step]
        lp = _lib.logprob_gather(outs.cuda()[:, -1], sampled[step].to(torch.int32).cuda()).cpu().numpy()
        assert (lp == host[:, step, 0]).all(), (step, lp, host[:, step, 0])
        # any other alignment: another summation order, the same value within the path's bar
        lp2 = _lib.logprob_gather(full[step].cuda(), sampled[step].to(torch.int32).cuda()).cpu().numpy()
        assert np.abs(lp2 - host[:, step, 0]).max() <= 2e-5


@pytest.mark.parametrize("attention", ["peaked", "flat"])
def test_thirty_two_ragged_streams_against_one_stream_each(attention):
    """What bench.py's `ragged_32_streams` leg asserts, as a test: 32 recordings of 5-30 s with transcripts of their own
    (2-9 segments, 40-160 tokens, whisper-base shapes) through ONE decoder loop, every fourth one also through
    transcribe() alone.
    peaked (cross-attention with a monotone ridge on the alignment heads, as a trained model has): north_star's bar for
        EVERY word -- times within 0.02 s, confidences within 1e-4, mean log-probabilities within 2e-4;
    flat (plain random init): same words, confidences and log-probabilities; word times within 0.02 s for at least 99 % of
        the words -- a batch of 32 and a batch of 1 round differently inside the backend's GEMMs, and a flat attention row
        turns that into a moved boundary where the script repeats a token (gates nothing in bench.py)."""
    import numpy as np
    import wordgaps as bench
    import many_helper as H
    import whisper_double as W
    from whisper_double.decoding import Script, set_row_scripts, set_script
    W.install()
    import whisper_timestamped as wt
    from whisper_timestamped import streams, words
    model = H.load_base("cuda:0", attention=attention)
    make_window = H.peaked_window if attention == "peaked" else H.ragged_window
    TS0, EOT = 50364, 50257
    g = torch.Generator().manual_seed(7)
    clips = [(0.05 * torch.randn(30 * 16000, generator=g)).float() for _ in range(4)]
    rs = np.random.RandomState(132)
    audios, wins = [], []
    for k in range(32):
        sec = float(rs.uniform(5.0, 30.0))
        audios.append(clips[k % 4][:int(sec * 16000)].clone())
        wins.append([make_window(rs, int(sec * 50), TS0, EOT)])
    scripts = [Script(w_) for w_ in wins]

    def on_group(idx):
        for i in idx:
            scripts[i].begin_window()
        set_row_scripts([scripts[i] for i in idx])
    words.RAW_CONFIDENCE = True
    streams.ON_GROUP_DECODE = on_group
    try:
        batch = wt.transcribe_batch(model, audios, max_streams=32, language="en", fp16=False)
        assert streams.LAST_RUN["decoder_loops"] == 1 and streams.LAST_RUN["streams_per_loop"] == [32]
        streams.ON_GROUP_DECODE = None
        set_row_scripts(None)
        worst = bench.NO_GAPS
        for k in range(0, 32, 4):
            set_script(Script(wins[k]))
            try:
                alone = wt.transcribe(model, audios[k], language="en", fp16=False)
            finally:
                set_script(None)
            worst = bench.merge_gaps(worst, bench.word_gaps(bench.words_of(batch[k]), bench.words_of(alone), f"recording {k}"))
    finally:
        words.RAW_CONFIDENCE = False
        streams.ON_GROUP_DECODE = None
        set_row_scripts(None)
    print(f"\n[{attention}] {bench.gaps_report(worst)}")
    ok = bench.gaps_ok(worst) if attention == "peaked" else bench.gaps_ok_between_batch_sizes(worst)
    assert worst[3] > 150 and ok, bench.gaps_report(worst)


@pytest.mark.gpu
def test_peaked_recordings_on_the_gpu_equal_the_cpu_reference_path(monkeypatch):
    """GPU vs the reference-shaped CPU path on the PEAKED double (bench.py's parity_vs_cpu_reference_path, as a test): four
    ragged recordings decoded together as 4 streams on the GPU, each also through transcribe() on the CPU with the
    oracle-backed kernels, unfused attention, a second projection per token and one alignment per segment (the reference's
    shape): every word within 0.02 s, confidences within 1e-4, mean log-probabilities within 2e-4."""
    import numpy as np
    import wordgaps as bench
    import many_helper as H
    import cpu_kernel_standin
    import whisper_double as W
    from whisper_double.decoding import Script, set_row_scripts, set_script
    W.install()
    import whisper_timestamped as wt
    from whisper_timestamped import efficient, streams, words
    model = H.load_base("cuda:0", attention="peaked")
    TS0, EOT = 50364, 50257
    g = torch.Generator().manual_seed(9)
    clip = (0.05 * torch.randn(30 * 16000, generator=g)).float()
    rs = np.random.RandomState(41)
    audios, wins = [], []
    for k in range(4):
        sec = float(rs.uniform(8.0, 30.0))
        audios.append(clip[:int(sec * 16000)].clone())
        wins.append([H.peaked_window(rs, int(sec * 50), TS0, EOT, lo=30, hi=90)])
    scripts = [Script(w_) for w_ in wins]

    def on_group(idx):
        for i in idx:
            scripts[i].begin_window()
        set_row_scripts([scripts[i] for i in idx])
    monkeypatch.setattr(words, "RAW_CONFIDENCE", True)
    streams.ON_GROUP_DECODE = on_group
    try:
        batch = wt.transcribe_batch(model, audios, max_streams=4, language="en", fp16=False)
    finally:
        streams.ON_GROUP_DECODE = None
        set_row_scripts(None)
    del model
    cpu_kernel_standin.install(monkeypatch)
    monkeypatch.setattr(efficient, "REUSE_DECODER_LOGITS", False)
    monkeypatch.setattr(efficient, "DEFER_ALIGNMENT", False)
    model_cpu = H.load_base("cpu", attention="peaked")
    worst = bench.NO_GAPS
    for k in range(4):
        set_script(Script(wins[k]))
        try:
            ref = wt.transcribe(model_cpu, audios[k], language="en", fp16=False)
        finally:
            set_script(None)
        worst = bench.merge_gaps(worst, bench.word_gaps(bench.words_of(batch[k]), bench.words_of(ref), f"recording {k} vs the CPU path"))
    print(f"\n{bench.gaps_report(worst)}")
    assert worst[3] > 60 and bench.gaps_ok(worst), bench.gaps_report(worst)
