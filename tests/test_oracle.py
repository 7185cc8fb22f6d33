"""CPU tests that PIN THE ORACLE: against fixtures produced by the reference's
own code (tests/golden/make_golden.py), against the reference's known-answer
test, and against independent implementations available in the image."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import align_ref as O
import synth
from golden.make_golden import build_case_inputs

G = os.path.join(os.path.dirname(__file__), "golden")
CASES = json.load(open(os.path.join(G, "align_cases.json"), encoding="utf-8"))
COSTS = np.load(os.path.join(G, "align_cost.npz"))


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_alignment_matches_reference_fixture(case):
    tokens, att, heads, mfcc, tok = build_case_inputs(case)
    words, internals = O.perform_word_alignment_ref(
        tokens, att, tok, use_space=case.get("use_space", True), mfcc=mfcc,
        refine_whisper_precision_nframes=case["refine"],
        remove_punctuation_from_words=case.get("remove_punct", False),
        alignment_heads=None if heads is None else np.array(heads),
        detect_disfluencies=case.get("disfl", False), return_internals=True,
        subwords_can_be_empty=not case.get("noempty", False))
    # the f64 cost built by the reference's own lines 1540-1568 (captured at the dtw call)
    ref_cost = COSTS[case["name"]].astype(np.float64)
    assert internals["cost"].shape == tuple(case["cost_shape"])
    assert np.array_equal(internals["cost"], ref_cost)
    assert internals["index1s"].tolist() == case["index1s"]
    assert internals["index2s"].tolist() == case["index2s"]
    got = [dict(text=w["text"], start=w["start"], end=w["end"], tokens=w["tokens"],
                tokens_indices=[int(x) for x in w["tokens_indices"]]) for w in words]
    assert got == case["words"]


def test_split_tokens_kat():
    kat = json.load(open(os.path.join(G, "split_tokens_kat.json"), encoding="utf-8"))
    for k in kat:
        tok = synth.StubTokenizer(multilingual=k["multilingual"])
        got = O.split_tokens_on_spaces_ref(list(k["tokens"]), tok)
        assert got == (k["words"], k["word_tokens"], k["word_tokens_indices"]), k["source"]


def test_find_start_padding_fixture():
    for p in json.load(open(os.path.join(G, "find_start_padding.json"))):
        rng = np.random.RandomState(p["seed"])
        m = rng.standard_normal((1, p["n_mels"], 3000)).astype(np.float32)
        if p["kind"] in ("zeros", "zero_col_inside"):
            m[..., p["col"]:] = 0.0
            if p["kind"] == "zero_col_inside":
                m[..., 1000] = 0.0
        elif p["kind"] == "allzero":
            m[:] = 0.0
        elif p["kind"] == "const_nonzero":
            m[..., p["col"]:] = 0.5
        assert O.find_start_padding_ref(torch.from_numpy(m)) == p["expected"], p


def test_dtw_optimal_cost_vs_exhaustive_recurrence():
    rng = np.random.RandomState(0)
    for _ in range(200):
        T, F = rng.randint(1, 8), rng.randint(1, 12)
        c = -rng.rand(T, F)
        r = O.dtw_ref(c, keep_internals=True)
        assert np.isclose(r.distance, O.dtw_bruteforce_cost(c), rtol=0, atol=1e-12)
        # path is monotone, single-step, closed ends, cost adds up exactly in path order
        assert (r.index1s[0], r.index2s[0]) == (0, 0) and (r.index1s[-1], r.index2s[-1]) == (T - 1, F - 1)
        d1, d2 = np.diff(r.index1s), np.diff(r.index2s)
        assert set(zip(d1.tolist(), d2.tolist())) <= {(1, 1), (0, 1), (1, 0)}
        acc = 0.0
        for i, j in zip(r.index1s, r.index2s):
            acc = acc + c[i, j] if (i, j) != (0, 0) else c[0, 0]
        assert acc == r.distance


def test_dtw_tie_order():
    """All-equal costs: every candidate ties -> pattern 1 (diagonal) must win
    where it exists, then pattern 2 (same token, previous frame): dtw-python's
    strict '<' first-wins argmin."""
    r = O.dtw_ref(np.zeros((3, 5)), keep_internals=True)
    sm = r.directionMatrix
    assert (sm[1:, 1:] == 1).all() and (sm[0, 1:] == 2).all() and (sm[1:, 0] == 3).all()
    assert r.index1s.tolist() == [0, 0, 0, 1, 2] and r.index2s.tolist() == [0, 1, 2, 3, 4]
    assert O.jumps_from_path(r.index1s, r.index2s).tolist() == [0, 3, 4, 4]


def test_dtw_vs_transformers_on_tie_free_inputs():
    gw = pytest.importorskip("transformers.models.whisper.generation_whisper")
    rng = np.random.RandomState(1)
    for _ in range(20):
        T, F = rng.randint(2, 30), rng.randint(30, 200)
        c = -rng.rand(T, F)
        c[0, 0] = c.min()
        r = O.dtw_ref(c)
        ti, tj = gw._dynamic_time_warping(c)
        assert np.array_equal(np.asarray(ti), r.index1s) and np.array_equal(np.asarray(tj), r.index2s)


def _tie_sets(rng, n, max_t=40, max_f=90):
    """Cost matrices on which the direction choice is decided by TIE ORDER: zeros, constants, quantised levels, the
    reference's pad-mask plateau (rows [:-1] zero from a column on), mixed sign."""
    out = [np.zeros((3, 5)), np.zeros((7, 7)), -np.ones((5, 11)), np.ones((4, 9))]
    for _ in range(n):
        T, F = int(rng.randint(1, max_t)), int(rng.randint(1, max_f))
        kind = rng.randint(4)
        if kind == 0:
            c = -(rng.randint(0, 4, size=(T, F)) / 4.0)
        elif kind == 1:
            c = rng.randint(-2, 3, size=(T, F)) / 2.0
        elif kind == 2:
            c = -rng.rand(T, F)
            if T > 1:
                c[:-1, int(F * rng.rand()):] = 0.0
        else:
            c = np.round(rng.standard_normal((T, F)), 1)
        if rng.rand() < 0.5:
            c[0, 0] = c.min()
        out.append(c.astype(np.float64))
    return out


def test_dtw_pattern_interpreter_equals_the_c_restatement():
    """oracle/dtw_patterns.py (generic loops over the pattern ROWS, as dtw-python's computeCM / backtrack run them) against
    oracle/dtw_ref.c (the two patterns hard-coded): same paths, same distances, bit for bit -- both step patterns, on
    tie-decided matrices.  The rows of the second pattern are the ones the reference passes at transcribe.py:1575-1580."""
    from oracle import dtw_patterns as P
    rng = np.random.RandomState(7)
    no_empty = P.StepPattern(P._c(1, 1, 1, -1,
                                  1, 0, 0, 1,
                                  2, 0, 1, -1,
                                  2, 0, 0, 1))
    for c in _tie_sets(rng, 120):
        for code, pat in ((0, P.symmetric1), (1, no_empty)):
            if code == 1 and c.shape[0] > c.shape[1]:
                with pytest.raises(ValueError):
                    P.dtw(c, step_pattern=pat)
                with pytest.raises(ValueError):
                    O.dtw_ref(c, step_pattern=1)
                continue
            a = P.dtw(c, step_pattern=pat)
            b = O.dtw_ref(c, step_pattern=code)
            assert np.array_equal(a.index1s, b.index1s) and np.array_equal(a.index2s, b.index2s), (code, c.shape)
            assert a.distance == b.distance


def test_dtw_pattern_interpreter_sees_a_swapped_tie_order():
    """Mutation: the same rows with patterns 2 and 3 exchanged (previous-token/same-frame tried before same-token/
    previous-frame) must give OTHER paths on tie-decided matrices -- i.e. the comparison above can tell the orders apart --
    and the same optimal distance (the set of moves is unchanged)."""
    from oracle import dtw_patterns as P
    swapped = P.StepPattern(P._c(1, 1, 1, -1, 1, 0, 0, 1,
                                 2, 1, 0, -1, 2, 0, 0, 1,
                                 3, 0, 1, -1, 3, 0, 0, 1))
    rng = np.random.RandomState(8)
    differ = 0
    sets = _tie_sets(rng, 60)
    for c in sets:
        a, b = P.dtw(c, step_pattern=P.symmetric1), P.dtw(c, step_pattern=swapped)
        assert np.isclose(a.distance, b.distance, rtol=0, atol=1e-9)
        differ += not (np.array_equal(a.index1s, b.index1s) and np.array_equal(a.index2s, b.index2s))
        ref = O.dtw_ref(c)
        assert np.array_equal(a.index1s, ref.index1s) and np.array_equal(a.index2s, ref.index2s)
    assert differ >= len(sets) // 4, differ             # (a third of the matrices of this seed: 22 of 64)


def test_dtw_pattern_interpreter_optimal_cost_vs_exhaustive_recurrence():
    from oracle import dtw_patterns as P
    rng = np.random.RandomState(9)
    for _ in range(100):
        T, F = rng.randint(1, 7), rng.randint(1, 10)
        c = -rng.rand(T, F)
        assert np.isclose(P.dtw(c).distance, O.dtw_bruteforce_cost(c), rtol=0, atol=1e-12)


def test_dtw_rejects_nan():
    c = np.zeros((3, 4)); c[1, 2] = np.nan
    with pytest.raises(ValueError):
        O.dtw_ref(c)


def test_median_filter_is_half_sample_symmetric():
    rng = np.random.RandomState(2)
    for F in (1, 2, 3, 4, 5, 9, 17, 200):
        x = rng.standard_normal((2, 3, F)).astype(np.float32)
        got = O.median_filter(x, (1, 1, 9))
        idx = np.arange(-4, F + 4)
        m = np.mod(idx, 2 * F)
        m = np.where(m < F, m, 2 * F - 1 - m)
        xp = x[..., m]
        want = np.stack([np.median(xp[..., k:k + 9], axis=-1) for k in range(F)], axis=-1)
        assert np.array_equal(got, want.astype(np.float32)), F


def test_mel_filters_vs_transformers():
    au = pytest.importorskip("transformers.audio_utils")
    for n in (80, 128):
        want = au.mel_filter_bank(201, n, 0.0, 8000.0, 16000, norm="slaney", mel_scale="slaney").T
        got = O.mel_filters_ref(n).numpy()
        assert np.allclose(got, want, rtol=1e-5, atol=1e-7)


def test_logmel_shapes_and_padding_detect():
    rng = np.random.RandomState(3)
    pcm = (0.1 * rng.standard_normal(16000 * 7)).astype(np.float32)
    mel = O.pad_or_trim_ref(O.log_mel_spectrogram_ref(torch.from_numpy(pcm)), 3000)
    assert mel.shape == (80, 3000)
    assert O.find_start_padding_ref(mel[None]) == 700


def test_confidence_rounding():
    lp = torch.tensor([-0.1, -0.2, -1.5])
    assert O.confidence_ref(lp) == round(float(np.exp(np.float32(-0.6))), 3)
    assert O.confidence_ref([]) == 0.0


def _peaky_cost(rng, T, F, quantised, density=1):
    """(T,F) cost rows with several bumps per token span: what find_peaks has to sort out."""
    x = np.zeros((T, F))
    for t in range(T):
        for _ in range(rng.randint(1, 10) * density):
            c, w, h = rng.uniform(0, F), rng.uniform(0.7, 5.0), rng.uniform(0.01, 0.3)
            x[t] += h * np.exp(-0.5 * ((np.arange(F) - c) / w) ** 2)
        x[t] += 0.004 * rng.rand(F)
    if quantised:                       # plateaus and exact ties
        x = np.round(x * 64) / 64
    return (-x).astype(np.float32)


def test_peak_restatement_vs_scipy():
    """The loop-by-loop restatement of scipy's find_peaks(width, prominence) (the text wt_peaks.hip follows) against
    scipy itself, as the reference calls it (transcribe.py:1663-1666)."""
    rng = np.random.RandomState(3)
    moved = 0
    for trial in range(80):
        T, F = rng.randint(1, 7), rng.randint(3, 200)
        cost = _peaky_cost(rng, T, F, quantised=trial % 3 == 0)
        cuts = np.sort(rng.randint(0, F, size=T - 1)) if T > 1 else np.zeros(0, dtype=np.int64)
        jumps = np.concatenate([[0], cuts, [F - 1]]).astype(np.int64)
        ref = O.jumps_start_ref(cost, jumps)
        got = O.jumps_start_restated(cost, jumps)
        assert np.array_equal(ref, got), (trial, ref, got)
        moved += int((ref != jumps).sum())
    assert moved > 20        # the inputs do exercise the "more than one peak" branch
