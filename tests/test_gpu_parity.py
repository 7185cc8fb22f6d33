"""GPU parity tests proper (-m gpu): the HIP path, called through the C ABI,
against the CPU oracle and the reference-generated fixtures.

Bars (BASELINE.json north_star): integer outputs (DTW path, jumps) bit-exact
for a given cost matrix; fp32 cost within 2e-6 relative of the oracle's torch
CPU arithmetic (1-ulp-class reduction-order noise); word start/end within
+-0.02 s (one 20 ms frame); confidences within 1e-4 before rounding.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import align_ref as O
import synth
from golden.make_golden import build_case_inputs

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


def _lib():
    from whisper_timestamped import _lib as L
    return L


def run_dtw(costs, want_path=True):
    """costs: list of (T,F) float32 numpy -> per-unit (jumps, path_i, path_j, dist)."""
    L = _lib()
    descs = L.make_descs(len(costs))
    for d, c in zip(descs, costs):
        d["T"], d["F"] = c.shape
        d["pad_from"] = -1
    n_cost, n_jumps, n_path = L.layout_outputs(descs)
    flat = np.zeros(n_cost, dtype=np.float32)
    for d, c in zip(descs, costs):
        flat[d["cost_offset"]:d["cost_offset"] + c.size] = c.ravel()
    cost = torch.from_numpy(flat).to(DEV)
    dd = L.descs_to_device(descs, DEV)
    jumps = torch.full((n_jumps,), -7, dtype=torch.int32, device=DEV)
    pi = torch.full((n_path,), -7, dtype=torch.int32, device=DEV)
    pj = torch.full((n_path,), -7, dtype=torch.int32, device=DEV)
    pl = torch.zeros(len(costs), dtype=torch.int32, device=DEV)
    dist = torch.zeros(len(costs), dtype=torch.float64, device=DEV)
    L.dtw_batch(cost, descs, dd, jumps, pi, pj, pl, dist)
    torch.cuda.synchronize()
    jumps, pi, pj, pl, dist = jumps.cpu().numpy(), pi.cpu().numpy(), pj.cpu().numpy(), pl.cpu().numpy(), dist.cpu().numpy()
    out = []
    for k, (d, c) in enumerate(zip(descs, costs)):
        T, F = c.shape
        n = int(pl[k])
        out.append((jumps[d["jumps_offset"]:d["jumps_offset"] + T + 1], pi[d["path_offset"]:d["path_offset"] + n],
                    pj[d["path_offset"]:d["path_offset"] + n], float(dist[k])))
    return out


def check_dtw_exact(costs):
    got = run_dtw(costs)
    for c, (jm, pi, pj, dist) in zip(costs, got):
        r = O.dtw_ref(c.astype(np.float64), keep_internals=True)
        assert np.array_equal(pi, r.index1s), f"path_i differs for shape {c.shape}"
        assert np.array_equal(pj, r.index2s), f"path_j differs for shape {c.shape}"
        assert np.array_equal(jm, O.jumps_from_path(r.index1s, r.index2s)), f"jumps differ for shape {c.shape}"
        assert dist == r.distance, (dist, r.distance)


def test_dtw_bit_exact_random_shapes():
    rng = np.random.RandomState(0)
    costs = []
    for _ in range(150):
        T = int(rng.choice([1, 2, 3, 7, 16, 33, 63, 64, 65, 100, 128, 129, 191, 200, 225, 256]))
        F = int(rng.choice([1, 2, 5, 15, 16, 17, 48, 49, 100, 144, 352, 700, 1500]))
        costs.append((-rng.rand(T, F)).astype(np.float32))
    check_dtw_exact(costs)


def test_dtw_bit_exact_ties_and_masks():
    """Exact ties everywhere (zeros, constant blocks, quantised costs): the
    direction choice must follow dtw-python's first-wins strict '<'."""
    rng = np.random.RandomState(1)
    costs = [np.zeros((5, 9), np.float32), np.zeros((70, 300), np.float32), -np.ones((130, 200), np.float32)]
    for _ in range(40):
        T, F = int(rng.randint(2, 230)), int(rng.randint(2, 600))
        c = -(rng.randint(0, 4, size=(T, F)) / 4.0).astype(np.float32)
        if rng.rand() < 0.5:
            c[:-1, int(F * rng.rand()):] = 0.0        # the reference's padding mask shape
        c[0, 0] = c.min()
        costs.append(c)
    check_dtw_exact(costs)


def test_dtw_bit_exact_positive_and_mixed_sign_costs():
    """include/wtalign.h promises dtw-python semantics for ANY finite cost, not only the product's (<= 0, cost[0,0] =
    min).  With cost[0,0] > 0 the first row is where a stale seed of cell (0,0) would show: g[0,j] must be the prefix
    sum of row 0."""
    rng = np.random.RandomState(21)
    costs = [np.ones((3, 7), np.float32), np.array([[2.0, 1.0, 1.0, 5.0], [1.0, 3.0, 0.5, 0.25]], np.float32),
             rng.rand(1, 40).astype(np.float32), rng.rand(70, 1).astype(np.float32)]
    for _ in range(60):
        T = int(rng.choice([1, 2, 5, 31, 64, 65, 130, 224, 256]))
        F = int(rng.choice([1, 3, 4, 33, 64, 65, 200, 611, 1500]))
        kind = rng.randint(3)
        c = rng.rand(T, F) if kind == 0 else rng.standard_normal((T, F)) if kind == 1 else rng.randint(-2, 3, size=(T, F)) / 2.0
        costs.append(c.astype(np.float32))
    check_dtw_exact(costs)
    # and the values themselves on one unit: the distance of a single-row unit is the plain sum of the row
    row = rng.rand(1, 300).astype(np.float32)
    (_, _, _, dist), = run_dtw([row])
    acc = 0.0
    for v in row[0]:
        acc += float(v)
    assert dist == acc


def test_dtw_kernel_equals_the_pattern_interpreter_on_tie_sets():
    """The HIP kernel against oracle/dtw_patterns.py -- the generic step-pattern interpreter that consumes the rows the
    reference builds (transcribe.py:1575-1580) and dtw-python's published symmetric1 rows -- on tie-decided matrices
    (zeros, quantised levels, pad-mask plateaus, mixed sign), both step patterns: paths, jumps and distances bit-exact.
    (A second restatement, structurally unlike oracle/dtw_ref.c, while dtw-python itself cannot be installed.)"""
    from oracle import dtw_patterns as P
    from test_oracle import _tie_sets
    L = _lib()
    rng = np.random.RandomState(31)
    costs = [c.astype(np.float32) for c in _tie_sets(rng, 80, max_t=70, max_f=160)]
    got = run_dtw(costs)
    for c, (jm, pi, pj, dist) in zip(costs, got):
        r = P.dtw(c.astype(np.float64))
        assert np.array_equal(pi, r.index1s) and np.array_equal(pj, r.index2s), c.shape
        assert np.array_equal(jm, O.jumps_from_path(r.index1s, r.index2s)) and dist == r.distance
    no_empty = P.StepPattern(P._c(1, 1, 1, -1, 1, 0, 0, 1, 2, 0, 1, -1, 2, 0, 0, 1))
    costs = [c for c in costs if c.shape[0] <= c.shape[1]]
    descs = L.make_descs(len(costs))
    for d, c in zip(descs, costs):
        d["T"], d["F"] = c.shape
        d["pad_from"] = -1
    n_cost, n_jumps, n_path = L.layout_outputs(descs)
    flat = np.zeros(n_cost, dtype=np.float32)
    for d, c in zip(descs, costs):
        flat[d["cost_offset"]:d["cost_offset"] + c.size] = c.ravel()
    jumps = torch.zeros(n_jumps, dtype=torch.int32, device=DEV)
    dist = torch.zeros(len(costs), dtype=torch.float64, device=DEV)
    L.dtw_batch(torch.from_numpy(flat).to(DEV), descs, L.descs_to_device(descs, DEV), jumps, dist=dist,
                step_pattern=L.WT_STEP_NO_EMPTY_SUBWORDS)
    jumps, dist = jumps.cpu().numpy(), dist.cpu().numpy()
    for k, (d, c) in enumerate(zip(descs, costs)):
        r = P.dtw(c.astype(np.float64), step_pattern=no_empty)
        j0 = int(d["jumps_offset"])
        assert np.array_equal(jumps[j0:j0 + c.shape[0] + 1], O.jumps_from_path(r.index1s, r.index2s)), c.shape
        assert dist[k] == r.distance


def test_dtw_bit_exact_max_sizes():
    rng = np.random.RandomState(2)
    costs = [(-rng.rand(256, 1500)).astype(np.float32), (-rng.rand(224, 1500)).astype(np.float32),
             (-rng.rand(256, 256)).astype(np.float32), (-rng.rand(64, 1792)).astype(np.float32),
             (-rng.rand(192, 1792)).astype(np.float32), (-rng.rand(256, 1760)).astype(np.float32)]
    check_dtw_exact(costs)


def test_dtw_rejects_unsupported():
    L = _lib()
    descs = L.make_descs(1)
    descs[0]["T"], descs[0]["F"] = 300, 100
    L.layout_outputs(descs)
    t = torch.zeros(300 * 100, device=DEV)
    with pytest.raises(L.WtError):
        L.dtw_batch(t, descs, L.descs_to_device(descs, DEV), torch.zeros(301, dtype=torch.int32, device=DEV))
    descs[0]["T"], descs[0]["F"] = 256, 1793      # one frame beyond WT_MAX_FRAMES
    L.layout_outputs(descs)
    t = torch.zeros(256 * 1793 + 4, device=DEV)
    with pytest.raises(L.WtError):
        L.dtw_batch(t, descs, L.descs_to_device(descs, DEV), torch.zeros(257, dtype=torch.int32, device=DEV))


def test_dtw_no_empty_subwords_pattern_bit_exact():
    """transcribe.py:1575-1580 (perform_word_alignment(subwords_can_be_empty=False)): symmetric1 without the previous-
    token/same-frame move -- against the oracle's restatement of dtw-python with that pattern; every token then owns at
    least one frame (strictly increasing jumps); T > F is refused like dtw-python's "no warping path"."""
    L = _lib()
    rng = np.random.RandomState(41)
    shapes = [(1, 1), (1, 9), (2, 2), (5, 5), (7, 100), (64, 64), (65, 300), (130, 131), (224, 1500), (256, 1792), (33, 40)]
    costs = [(-rng.rand(T, F)).astype(np.float32) for T, F in shapes]
    costs += [np.zeros((12, 40), np.float32), -(rng.randint(0, 3, size=(90, 400)) / 2.0).astype(np.float32)]
    descs = L.make_descs(len(costs))
    for d, c in zip(descs, costs):
        d["T"], d["F"] = c.shape
        d["pad_from"] = -1
    n_cost, n_jumps, n_path = L.layout_outputs(descs)
    flat = np.zeros(n_cost, dtype=np.float32)
    for d, c in zip(descs, costs):
        flat[d["cost_offset"]:d["cost_offset"] + c.size] = c.ravel()
    jumps = torch.full((n_jumps,), -7, dtype=torch.int32, device=DEV)
    pi = torch.full((n_path,), -7, dtype=torch.int32, device=DEV)
    pj = torch.full((n_path,), -7, dtype=torch.int32, device=DEV)
    pl = torch.zeros(len(costs), dtype=torch.int32, device=DEV)
    dist = torch.zeros(len(costs), dtype=torch.float64, device=DEV)
    L.dtw_batch(torch.from_numpy(flat).to(DEV), descs, L.descs_to_device(descs, DEV), jumps, pi, pj, pl, dist,
                step_pattern=L.WT_STEP_NO_EMPTY_SUBWORDS)
    torch.cuda.synchronize()
    j, pi, pj, pl, dist = jumps.cpu().numpy(), pi.cpu().numpy(), pj.cpu().numpy(), pl.cpu().numpy(), dist.cpu().numpy()
    for k, (d, c) in enumerate(zip(descs, costs)):
        T, F = c.shape
        r = O.dtw_ref(c.astype(np.float64), step_pattern=1)
        n = int(pl[k])
        assert n == len(r.index1s) == F
        assert np.array_equal(pi[d["path_offset"]:d["path_offset"] + n], r.index1s), c.shape
        assert np.array_equal(pj[d["path_offset"]:d["path_offset"] + n], r.index2s), c.shape
        jm = j[d["jumps_offset"]:d["jumps_offset"] + T + 1]
        assert np.array_equal(jm, O.jumps_from_path(r.index1s, r.index2s)) and (np.diff(jm[:-1]) > 0).all()
        assert dist[k] == r.distance
    descs = L.make_descs(1)
    descs[0]["T"], descs[0]["F"] = 9, 8
    L.layout_outputs(descs)
    with pytest.raises(L.WtError, match="no warping path"):
        L.dtw_batch(torch.zeros(9 * 8 + 4, device=DEV), descs, L.descs_to_device(descs, DEV),
                    torch.zeros(10, dtype=torch.int32, device=DEV), step_pattern=L.WT_STEP_NO_EMPTY_SUBWORDS)


def test_dtw_largest_shape_is_supported():
    """T = 256 with F = 1792: refused in round 1 (direction planes in LDS: > 160 KiB); the planes live in the scratch
    arena now, so the whole (WT_MAX_TOKENS, WT_MAX_FRAMES) range runs -- several units per launch class."""
    rng = np.random.RandomState(31)
    check_dtw_exact([(-rng.rand(256, 1792)).astype(np.float32), (-rng.rand(193, 1790)).astype(np.float32),
                     (-rng.rand(255, 1761)).astype(np.float32), (-rng.rand(3, 1792)).astype(np.float32)])


# ---------------------------------------------------------------------------
def run_cost(qk_list, heads_list, windows, pads, dtype=torch.float32, medfilt_width=9):
    """qk_list[k]: (L*H, T, 1500) numpy; heads_list: flat head indices (shared); windows[k]=(start,end)."""
    L = _lib()
    n = len(qk_list)
    descs = L.make_descs(n)
    off = 0
    for d, q, (s, e), p in zip(descs, qk_list, windows, pads):
        d["qk_offset"] = off
        d["head_stride"] = q.shape[1] * q.shape[2]
        d["row_stride"] = q.shape[2]
        d["T"], d["F"], d["start_token"], d["pad_from"] = q.shape[1], e - s, s, p
        off += q.size
    n_cost, n_jumps, n_path = L.layout_outputs(descs)
    qk = torch.from_numpy(np.concatenate([q.ravel() for q in qk_list])).to(DEV).to(dtype)
    hi = torch.tensor(heads_list, dtype=torch.int32, device=DEV)
    cost = torch.full((n_cost,), float("nan"), dtype=torch.float32, device=DEV)
    dd = L.descs_to_device(descs, DEV)
    L.cost_batch(qk, descs, dd, hi, cost, medfilt_width=medfilt_width)
    torch.cuda.synchronize()
    c = cost.cpu().numpy()
    return [c[d["cost_offset"]:d["cost_offset"] + d["T"] * d["F"]].reshape(d["T"], d["F"]) for d in descs]


def oracle_cost(q, heads, window, pad, medfilt_width=9):
    s, e = window
    sel = torch.from_numpy(q[heads][:, :, s:e])
    md = pad if pad >= 0 else None
    return O.cost_matrix_ref(sel, medfilt_width, 1.0, md, 0 if md else 0)


def test_cost_matches_oracle():
    rng = np.random.RandomState(3)
    qk_list, windows, pads = [], [], []
    heads = [25, 34, 35, 39, 41, 42, 44, 46]          # whisper-base alignment heads, flat l*8+h
    shapes = [(2, 0, 3), (3, 10, 14), (5, 0, 9), (8, 100, 245), (18, 275, 523), (11, 0, 256), (11, 1, 258),
              (40, 200, 968), (30, 0, 769), (60, 0, 1280), (33, 219, 1500), (64, 0, 1500), (10, 700, 1400)]
    for k, (T, s, e) in enumerate(shapes):
        qk_list.append(synth.synth_qk(100 + k, 48, T, lo=s, hi=e))
        windows.append((s, e))
        pads.append(-1 if k % 3 else max((e - s) // 2, 1))
    got = run_cost(qk_list, heads, windows, pads)
    worst = 0.0
    for q, w, p, g in zip(qk_list, windows, pads, got):
        ref = oracle_cost(q, heads, w, p)
        assert g.shape == ref.shape and np.isfinite(g).all()
        err = np.abs(g.astype(np.float64) - ref).max() / np.abs(ref).max()
        worst = max(worst, err)
        assert err < 2e-6, (w, err)
        if p >= 0:
            assert (g[:-1, p:] == 0).all() or (p == 0)
        assert g[0, 0] == g.min()
    print(f"max relative cost error vs oracle: {worst:.3e}")


@pytest.mark.parametrize("width", [1, 3, 5, 7])
def test_cost_other_median_widths_match_the_oracle(width):
    """medfilt_width is a parameter of the seam (transcribe.py:1439; scipy.ndimage.median_filter(w, (1, 1, width)), the
    reference's default and only caller value being 9): the odd widths below 9 against the oracle's scipy call, every F
    class, windows shorter than the filter, fp32 and fp16 rows; even widths and widths above 9 are refused."""
    L = _lib()
    heads = [1, 3, 4, 7, 9, 10]
    shapes = [(2, 0, 3), (3, 10, 14), (4, 5, 7), (8, 100, 245), (18, 275, 523), (11, 1, 258), (30, 0, 769), (9, 0, 1025),
              (20, 0, 1281), (33, 219, 1500), (5, 0, 1)]
    qs = [synth.synth_qk(500 + k, 12, T, lo=s, hi=e) for k, (T, s, e) in enumerate(shapes)]
    windows = [(s, e) for _, s, e in shapes]
    pads = [-1 if k % 2 else max((e - s) // 2, 1) for k, (_, s, e) in enumerate(shapes)]
    for q, w, p, g in zip(qs, windows, pads, run_cost(qs, heads, windows, pads, medfilt_width=width)):
        ref = oracle_cost(q, heads, w, p, medfilt_width=width)
        assert np.abs(g.astype(np.float64) - ref).max() / np.abs(ref).max() < 2e-6, (width, w)
        assert g[0, 0] == g.min()
    qh = [q.astype(np.float16).astype(np.float32) for q in qs]
    for q, w, p, g in zip(qh, windows, pads, run_cost(qh, heads, windows, pads, dtype=torch.float16, medfilt_width=width)):
        ref = oracle_cost(q, heads, w, p, medfilt_width=width)
        assert np.abs(g.astype(np.float64) - ref).max() / np.abs(ref).max() < 2e-6, (width, w, "fp16")
    if width == 1:
        for bad in (0, 2, 8, 11):
            with pytest.raises(L.WtError, match="medfilt_width"):
                run_cost(qs[:1], heads, windows[:1], pads[:1], medfilt_width=bad)
        # the seam itself: perform_word_alignment(medfilt_width=5) == the oracle's
        import whisper_timestamped as wt
        tok = synth.StubTokenizer()
        tokens = synth.synth_segment_tokens(9, 12, 100, 380, tok)
        qk = synth.synth_qk(9, 48, len(tokens), lo=75, hi=405).reshape(6, 1, 8, len(tokens), 1500)
        att = [torch.from_numpy(qk[l]) for l in range(6)]
        hp = np.array([(3, 1), (4, 2), (5, 4)])
        got = wt.perform_word_alignment(tokens, [a.to(DEV) for a in att], tok, alignment_heads=hp, medfilt_width=5,
                                        refine_whisper_precision_nframes=25, detect_disfluencies=False)
        want = O.perform_word_alignment_ref(tokens, att, tok, alignment_heads=hp, medfilt_width=5,
                                            refine_whisper_precision_nframes=25, detect_disfluencies=False)
        assert [(w["text"], w["start"], w["end"]) for w in got] == [(w["text"], w["start"], w["end"]) for w in want]


def test_cost_fp16_input_close_to_fp32_oracle():
    """fp16 QK storage is a build-side extension (the reference only sees fp32):
    quantify against the fp32 oracle on the same (fp16-rounded) logits.  Even / odd window starts exercise the
    paired (4-byte aligned) and the element-wise row loads, odd F the unpaired last element."""
    heads = list(range(6))
    shapes = [(20, 100, 400), (7, 101, 400), (9, 100, 401), (5, 3, 8), (3, 2, 5), (4, 0, 2), (33, 0, 1500), (12, 11, 1500)]
    qs = [synth.synth_qk(7 + k, 6, T, lo=s, hi=e).astype(np.float16).astype(np.float32) for k, (T, s, e) in enumerate(shapes)]
    got = run_cost(qs, heads, [(s, e) for _, s, e in shapes], [-1] * len(shapes), dtype=torch.float16)
    for q, (T, s, e), g in zip(qs, shapes, got):
        ref = oracle_cost(q, heads, (s, e), -1)
        assert np.abs(g - ref).max() / np.abs(ref).max() < 2e-6, (T, s, e)


@pytest.mark.parametrize("case", json.load(open(os.path.join(G, "align_cases.json"), encoding="utf-8")),
                         ids=lambda c: c["name"])
def test_perform_word_alignment_matches_reference_fixture(case):
    """The drop-in perform_word_alignment on the GPU vs the words the reference's
    own code produced (tests/golden/make_golden.py)."""
    import whisper_timestamped as wt
    tokens, att, heads, mfcc, tok = build_case_inputs(case)
    ah = None if heads is None else np.array(heads)
    words = wt.perform_word_alignment(
        tokens, [a.to(DEV) for a in att], tok, use_space=case.get("use_space", True),
        mfcc=None if mfcc is None else mfcc.to(DEV), refine_whisper_precision_nframes=case["refine"],
        remove_punctuation_from_words=case.get("remove_punct", False), alignment_heads=ah,
        detect_disfluencies=case.get("disfl", False), subwords_can_be_empty=not case.get("noempty", False))
    exp = case["words"]
    assert [w["text"] for w in words] == [w["text"] for w in exp]
    assert [w["tokens"] for w in words] == [w["tokens"] for w in exp]
    assert [[int(x) for x in w["tokens_indices"]] for w in words] == [w["tokens_indices"] for w in exp]
    dt = max([0.0] + [max(abs(a["start"] - b["start"]), abs(a["end"] - b["end"])) for a, b in zip(words, exp)])
    assert dt <= 0.02 + 1e-9, dt


def test_batch_of_real_shapes_vs_oracle():
    """160 units with the T/F mix measured on the reference goldens, one
    launch set; jumps compared with the oracle run on the oracle's own cost."""
    import whisper_timestamped as wt
    from whisper_timestamped.alignment import AlignmentBatch, AlignmentUnit
    tok = synth.StubTokenizer()
    Ts, Fs = synth.draw_real_shapes(5, 160)
    heads = [25, 34, 35, 39, 41, 42, 44, 46]
    batch = AlignmentBatch(keep_cost=True)
    refs = []
    rng = np.random.RandomState(9)
    for k, (T, F) in enumerate(zip(Ts, Fs)):
        T, F = int(T), int(F)
        s = int(rng.randint(0, 1500 - F + 1))
        q = synth.synth_qk(1000 + k, 48, T, lo=s, hi=s + F)
        sel = torch.from_numpy(q[heads])
        pad = -1 if k % 16 else int(rng.randint(F // 2, F))
        refs.append(O.cost_matrix_ref(sel[:, :, s:s + F], 9, 1.0, pad if pad >= 0 else None, 0))
        batch.add(AlignmentUnit(tokens=[0] * T, qk=sel.to(DEV).contiguous(), start_token=s, end_token=s + F, pad_from=pad,
                                words=[], word_pieces=[], word_ids=[], punct_counts=[], refine_nframes=25,
                                unfinished_decoding=False, detect_disfluencies=False, tokenizer=tok))
    # run only the device part
    import whisper_timestamped.alignment as A
    orig = A.finish_unit
    A.finish_unit = lambda u, j, c=None: j
    try:
        jumps = batch.run()
    finally:
        A.finish_unit = orig
    worst = 0
    for k, (jm, ref) in enumerate(zip(jumps, refs)):
        r = O.dtw_ref(ref)
        want = O.jumps_from_path(r.index1s, r.index2s)
        worst = max(worst, int(np.abs(jm - want).max()))
        # and bit-exact when the oracle DTW is fed the GPU's own cost
        g = batch.unit_cost(k).cpu().numpy().astype(np.float64)
        r2 = O.dtw_ref(g)
        assert np.array_equal(jm, O.jumps_from_path(r2.index1s, r2.index2s))
    print(f"max |jump difference| vs oracle end-to-end over 160 units: {worst} frame(s)")
    # how many units differ at all, and by how much, goes on record next to the profiles (profiles/*_parity_units.json)
    n_diff = 0
    for jm, ref in zip(jumps, refs):
        r = O.dtw_ref(ref)
        n_diff += int(np.any(jm != O.jumps_from_path(r.index1s, r.index2s)))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_units.json"), "w") as f:
        json.dump({"test": "test_batch_of_real_shapes_vs_oracle", "units": len(refs), "units_with_jump_diff": n_diff,
                   "worst_abs_jump_diff_frames": worst,
                   "note": "GPU cost (fp32, own reduction order) -> GPU DTW vs oracle cost -> oracle DTW; with the GPU's own "
                           "cost fed to the oracle DTW all units are bit-identical (asserted)"}, f)
    assert worst <= 1


def _align_units(shapes, dtype, flags, seed=0, n_heads=8, want_path=True):
    """cost, jumps, path, distance of a batch of (T, start, end) units through wt_align_batch_v3 with `flags`."""
    L = _lib()
    qk_list = [synth.synth_qk(seed + k, n_heads, T, lo=s, hi=e) for k, (T, s, e) in enumerate(shapes)]
    order = L.launch_order([(T, e - s) for T, s, e in shapes])
    descs = L.make_descs(len(shapes))
    offs, off = [], 0
    for q in qk_list:
        offs.append(off)
        off += q.size
    for d, i in zip(descs, order):
        T, s, e = shapes[i]
        q = qk_list[i]
        d["qk_offset"], d["head_stride"], d["row_stride"] = offs[i], q.shape[1] * q.shape[2], q.shape[2]
        d["T"], d["F"], d["start_token"], d["pad_from"] = T, e - s, s, (-1 if i % 3 else max((e - s) // 2, 1))
    n_cost, n_jumps, n_path = L.layout_outputs(descs)
    qk = torch.from_numpy(np.concatenate([q.ravel() for q in qk_list])).to(DEV).to(dtype)
    out = dict(descs=descs, order=order, qk_list=qk_list,
               cost=torch.full((n_cost,), float("nan"), dtype=torch.float32, device=DEV),
               jumps=torch.full((n_jumps,), -7, dtype=torch.int32, device=DEV),
               pi=torch.full((n_path,), -7, dtype=torch.int32, device=DEV), pj=torch.full((n_path,), -7, dtype=torch.int32, device=DEV),
               pl=torch.zeros(len(shapes), dtype=torch.int32, device=DEV), dist=torch.zeros(len(shapes), dtype=torch.float64, device=DEV))
    L.align_batch(qk, descs, L.descs_to_device(descs, DEV), torch.arange(n_heads, dtype=torch.int32, device=DEV), out["cost"],
                  out["jumps"], *( (out["pi"], out["pj"], out["pl"], out["dist"]) if want_path else (None, None, None, None)),
                  flags=flags)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_fused_small_units_equal_the_batched_kernels(dtype):
    """wt_align_batch_v3: units whose (T, F) matrix fits a workgroup's LDS (the reference's per-segment shape) leave the
    batched kernels after the row pass: ONE fused kernel (wt_small.hip) does column norm, cost[0,0], DTW and backtrack.
    cost, jumps, path and distance must be BIT-identical to the batched kernels' (colnorm / fix00 / dtw) on the same
    units -- every F class, tiny units, odd F, pad masks, one to four sweeping waves, the largest qualifying shapes --
    and the jumps bit-exact against the oracle DTW fed with that cost."""
    L = _lib()
    rng = np.random.RandomState(31)
    shapes = [(1, 0, 1), (1, 5, 9), (2, 0, 3), (3, 10, 14), (5, 1, 8), (9, 100, 245), (11, 3, 147), (17, 275, 523), (11, 3, 258),
              (64, 0, 256), (64, 1000, 1447), (63, 7, 300), (31, 1, 770), (30, 0, 1025), (16, 0, 1500), (4, 200, 1197), (8, 0, 1792 - 300),
              (33, 2, 900), (65, 0, 200), (224, 0, 1500), (12, 0, 64), (2, 1400, 1500), (74, 100, 400), (128, 0, 190), (129, 5, 160),
              (200, 0, 100), (256, 0, 60), (192, 3, 130), (100, 0, 260), (70, 1, 2), (130, 0, 1), (130, 0, 60), (150, 2, 32), (191, 0, 9)]
    Ts, Fs = synth.draw_real_shapes(77, 60)
    for T, F in zip(Ts, Fs):
        s = int(rng.randint(0, 1500 - int(F) + 1))
        shapes.append((int(T), s, s + int(F)))
    fused = _align_units(shapes, dtype, L.WT_ALIGN_KEEP_COST)
    plain = _align_units(shapes, dtype, L.WT_ALIGN_KEEP_COST | L.WT_ALIGN_NO_FUSED_SMALL_UNITS)
    n_small = 0
    for k, d in enumerate(fused["descs"]):
        T, F = int(d["T"]), int(d["F"])
        c0, j0, p0 = int(d["cost_offset"]), int(d["jumps_offset"]), int(d["path_offset"])
        a, b = fused["cost"][c0:c0 + T * F], plain["cost"][c0:c0 + T * F]
        assert torch.equal(a, b), f"cost of unit (T={T}, F={F}) differs between the fused and the batched kernels"
        assert torch.equal(fused["jumps"][j0:j0 + T + 1], plain["jumps"][j0:j0 + T + 1]), (T, F)
        n = int(plain["pl"][k])
        assert int(fused["pl"][k]) == n and torch.equal(fused["pi"][p0:p0 + n], plain["pi"][p0:p0 + n]) and \
            torch.equal(fused["pj"][p0:p0 + n], plain["pj"][p0:p0 + n]), (T, F)
        assert float(fused["dist"][k]) == float(plain["dist"][k]), (T, F)
        r = O.dtw_ref(a.reshape(T, F).cpu().numpy().astype(np.float64))
        assert np.array_equal(fused["jumps"][j0:j0 + T + 1].cpu().numpy(), O.jumps_from_path(r.index1s, r.index2s)), (T, F)
        n_small += int(L.small_unit(T, F))
    assert n_small > 60 and n_small < len(shapes)
    # the row pass as one launch per F class (what large batches get) instead of the small batch's single launch: same bits
    per_class = _align_units(shapes, dtype, L.WT_ALIGN_KEEP_COST | L.WT_ALIGN_ROWS_PER_CLASS)
    for d in per_class["descs"]:
        c0, n = int(d["cost_offset"]), int(d["T"]) * int(d["F"])
        assert torch.equal(per_class["cost"][c0:c0 + n], fused["cost"][c0:c0 + n]), (int(d["T"]), int(d["F"]))
    assert torch.equal(per_class["jumps"], fused["jumps"])
    # without WT_ALIGN_KEEP_COST the fused units leave no matrix behind, the others do; jumps are the same
    lean = _align_units(shapes, dtype, 0, want_path=False)
    assert torch.equal(lean["jumps"], plain["jumps"])
    kept = [not L.small_unit(int(d["T"]), int(d["F"])) for d in lean["descs"]]
    for d, k in zip(lean["descs"], kept):
        c0, n = int(d["cost_offset"]), int(d["T"]) * int(d["F"])
        if k:
            assert torch.equal(lean["cost"][c0:c0 + n], plain["cost"][c0:c0 + n])
    assert any(kept) and not all(kept)


def test_full_size_batch_properties():
    """BASELINE configs[1] size: 32 units of (8 heads, 224 tokens, 1500 frames).
    Size-independent properties + spot parity on 2 units."""
    L = _lib()
    n, T, F, A = 32, 224, 1500, 8
    g = torch.Generator(device="cpu").manual_seed(1234)
    qk = torch.randn((n, A, T, F), generator=g, dtype=torch.float32)
    for b in range(n):
        stairs = np.sort(np.random.RandomState(b).randint(0, F, size=T))
        for t in range(T):
            qk[b, :, t, max(stairs[t] - 1, 0):stairs[t] + 2] += 6.0
    qk_d = qk.to(DEV)
    descs = L.make_descs(n)
    for b, d in enumerate(descs):
        d["qk_offset"], d["head_stride"], d["row_stride"] = b * A * T * F, T * F, F
        d["T"], d["F"], d["start_token"], d["pad_from"] = T, F, 0, -1
    n_cost, n_jumps, n_path = L.layout_outputs(descs)
    dd = L.descs_to_device(descs, DEV)
    cost = torch.empty(n_cost, device=DEV)
    jumps = torch.empty(n_jumps, dtype=torch.int32, device=DEV)
    pi = torch.empty(n_path, dtype=torch.int32, device=DEV)
    pj = torch.empty(n_path, dtype=torch.int32, device=DEV)
    pl = torch.empty(n, dtype=torch.int32, device=DEV)
    dist = torch.empty(n, dtype=torch.float64, device=DEV)
    L.align_batch(qk_d, descs, dd, torch.arange(A, dtype=torch.int32, device=DEV), cost, jumps, pi, pj, pl, dist)
    torch.cuda.synchronize()
    cost_h, jm, pi, pj, pl, dist = cost.cpu().numpy(), jumps.cpu().numpy(), pi.cpu().numpy(), pj.cpu().numpy(), \
        pl.cpu().numpy(), dist.cpu().numpy()
    for b, d in enumerate(descs):
        c = cost_h[d["cost_offset"]:d["cost_offset"] + T * F].reshape(T, F)
        j = jm[d["jumps_offset"]:d["jumps_offset"] + T + 1]
        assert j[0] == 0 and j[-1] == F - 1 and (np.diff(j) >= 0).all()
        # columns of -cost have unit L2 norm (except the overwritten [0,0])
        nrm = np.sqrt((c.astype(np.float64)[:, 1:] ** 2).sum(0))
        assert np.abs(nrm - 1).max() < 1e-5
        p1, p2 = pi[d["path_offset"]:d["path_offset"] + pl[b]], pj[d["path_offset"]:d["path_offset"] + pl[b]]
        assert (p1[0], p2[0]) == (0, 0) and (p1[-1], p2[-1]) == (T - 1, F - 1)
        steps = set(zip(np.diff(p1).tolist(), np.diff(p2).tolist()))
        assert steps <= {(1, 1), (0, 1), (1, 0)}
        acc = 0.0
        for a, bb in zip(p1, p2):
            acc = float(c[a, bb]) if (a, bb) == (0, 0) else acc + float(c[a, bb])
        assert acc == dist[b]
        # the ridge is recovered: most tokens within a few frames of their staircase position
        stairs = np.sort(np.random.RandomState(b).randint(0, F, size=T))
        assert np.median(np.abs(j[:-1] - stairs)) <= 3
    for b in (0, 31):
        d = descs[b]
        ref = O.cost_matrix_ref(qk[b], 9, 1.0, None, 0)
        c = cost_h[d["cost_offset"]:d["cost_offset"] + T * F].reshape(T, F)
        assert np.abs(c - ref).max() / np.abs(ref).max() < 2e-6
        r = O.dtw_ref(c.astype(np.float64))
        assert np.array_equal(jm[d["jumps_offset"]:d["jumps_offset"] + T + 1], O.jumps_from_path(r.index1s, r.index2s))


# ---------------------------------------------------------------------------
def test_logprob_gather_vs_oracle():
    L = _lib()
    rng = np.random.RandomState(4)
    for V, n in [(51865, 40), (51864, 7), (51866, 5), (1000, 3), (7, 2)]:
        logits = (rng.standard_normal((n, V)) * 4).astype(np.float32)
        toks = rng.randint(0, V, size=n).astype(np.int32)
        got = L.logprob_gather(torch.from_numpy(logits).to(DEV), torch.from_numpy(toks)).cpu()
        want = O.token_logprob_gather_ref(torch.from_numpy(logits), toks)
        assert (got - want).abs().max() < 2e-5, (V, (got - want).abs().max())
        # with a suppression mask (the logit filters' -inf), shared and per-row
        mask = rng.rand(V) < 0.3
        mask[toks] = False
        got = L.logprob_gather(torch.from_numpy(logits).to(DEV), torch.from_numpy(toks), torch.from_numpy(mask)).cpu()
        want = O.token_logprob_gather_ref(torch.from_numpy(logits), toks, np.broadcast_to(mask, (n, V)).copy())
        assert (got - want).abs().max() < 2e-5
        maskn = rng.rand(n, V) < 0.5
        maskn[np.arange(n), toks] = False
        got = L.logprob_gather(torch.from_numpy(logits).to(DEV), torch.from_numpy(toks), torch.from_numpy(maskn)).cpu()
        want = O.token_logprob_gather_ref(torch.from_numpy(logits), toks, maskn)
        assert (got - want).abs().max() < 2e-5
    # confidence = exp(mean(logprobs)) within 1e-4 before rounding
    lp_g, lp_o = got.numpy(), want.numpy()
    assert abs(np.exp(lp_g.mean()) - O.confidence_raw_ref(lp_o)) < 1e-4


def test_logprob_gather_rows_vs_oracle():
    """wt_logprob_gather_rows: an explicit (row, token) list over a padded (B * T_max, V) block -- rows repeated,
    skipped, out of order -- equals the oracle's log_softmax(...)[row, token], and wt_logprob_gather_batch on a
    copy of the same rows (up to summation order: the 16-byte alignment of a row decides its head/body/tail split)."""
    L = _lib()
    rng = np.random.RandomState(8)
    for V, n_rows, dtype in [(51865, 96, torch.float32), (51864, 17, torch.float32), (51866, 33, torch.float16)]:
        logits = torch.from_numpy((rng.standard_normal((n_rows, V)) * 4).astype(np.float32)).to(dtype)
        idx = rng.randint(0, n_rows, size=150).astype(np.int32)
        idx[:5] = [n_rows - 1, 0, 0, 7, n_rows - 1]
        toks = rng.randint(0, V, size=idx.size).astype(np.int32)
        dl = logits.to(DEV)
        got = L.logprob_gather_rows(dl, torch.from_numpy(idx).to(DEV), torch.from_numpy(toks).to(DEV)).cpu()
        want = O.token_logprob_gather_ref(logits.float()[idx.astype(np.int64)], toks)
        assert (got - want).abs().max() < 2e-5, (V, (got - want).abs().max())
        same = L.logprob_gather(dl[torch.from_numpy(idx).long().to(DEV)].contiguous(), torch.from_numpy(toks)).cpu()
        assert (got - same).abs().max() < 1e-5
    out = torch.full((10,), 7.0, device=DEV)
    L.logprob_gather_rows(dl, torch.zeros(4, dtype=torch.int32, device=DEV), torch.zeros(4, dtype=torch.int32, device=DEV), out=out)
    assert (out[4:] == 7.0).all() and torch.isfinite(out[:4]).all()


def test_qk_rows_batch_vs_torch_and_single_window():
    """wt_qk_rows_batch (every window, every hooked layer, one launch) == (q * s) @ (k * s)^T in fp32 torch for the
    selected heads, == wt_qk_rows window by window; rows outside [row_begin, row_end) are not touched."""
    from whisper_timestamped.capture import QKCaptureRing
    L = _lib()
    g = torch.Generator().manual_seed(12)
    H, hd, n_ctx, B, n_q = 6, 64, 1500, 3, 37
    D = H * hd
    pairs = [(0, 1), (1, 0), (1, 5), (2, 3), (2, 4)]
    sel_layer = torch.tensor([p[0] for p in pairs], dtype=torch.int32, device=DEV)
    sel_head = torch.tensor([p[1] for p in pairs], dtype=torch.int32, device=DEV)
    sel_slot = torch.arange(len(pairs), dtype=torch.int32, device=DEV)
    for dtype, tol in ((torch.float32, 2e-5), (torch.float16, 2e-2)):
        qs = [(torch.randn((B, n_q, D), generator=g) * 0.7).to(dtype).to(DEV) for _ in range(3)]
        ks = [(torch.randn((B, n_ctx, D), generator=g) * 0.7).to(dtype).to(DEV) for _ in range(3)]
        ring = torch.full((B, len(pairs), n_q + 2, n_ctx), -77.0, device=DEV)
        lo = torch.tensor([3, 0, 36], dtype=torch.int32, device=DEV)
        hi = torch.tensor([37, 20, 37], dtype=torch.int32, device=DEV)
        L.qk_rows_batch(qs, ks, sel_layer, sel_head, sel_slot, ring, row_begin=lo, row_end=hi, ring_row0=1)
        torch.cuda.synchronize()
        scale = hd ** -0.25
        for b in range(B):
            single = QKCaptureRing(DEV, pairs, n_hooked_layers=3, n_heads=H, n_ctx=n_ctx, capacity=n_q + 2)
            for layer in range(3):
                single.write_from_projections(layer, qs[layer][b:b + 1], ks[layer][b:b + 1], row0=1, n_rows=n_q)
            for slot, (l, h) in enumerate(pairs):
                qh = (qs[l][b].float() * scale)[:, h * hd:(h + 1) * hd]
                kh = (ks[l][b].float() * scale)[:, h * hd:(h + 1) * hd]
                want = qh @ kh.T                                            # (n_q, n_ctx)
                a, e = int(lo[b]), int(hi[b])
                got = ring[b, slot, 1 + a:1 + e]
                assert (got - want[a:e]).abs().max().item() <= tol
                gap = (got - single.buf[slot, 1 + a:1 + e]).abs().max().item()      # same fma chain, another instruction mix
                assert gap <= 1e-5 if dtype == torch.float32 else gap <= 2e-2, gap
                assert (ring[b, slot, :1 + a] == -77.0).all() and (ring[b, slot, 1 + e:] == -77.0).all()
    # fp16 ring (storage option)
    ring16 = torch.zeros((B, len(pairs), n_q, n_ctx), dtype=torch.float16, device=DEV)
    L.qk_rows_batch(qs, ks, sel_layer, sel_head, sel_slot, ring16)
    ring32 = torch.zeros((B, len(pairs), n_q, n_ctx), dtype=torch.float32, device=DEV)
    L.qk_rows_batch(qs, ks, sel_layer, sel_head, sel_slot, ring32)
    assert torch.equal(ring16, ring32.half())


def test_qk_rows_streams_writes_each_streams_block_and_nothing_else():
    """wt_qk_rows_streams (B decoder streams, ONE launch per decoded token): batch entry b writes the LAST query row of every
    selected head into ring block ring_index[b] at the given row -- the streams of a call being any subset of the blocks,
    in any order; == (q * s) @ (k * s)^T in fp32 torch, == the single-stream entry; every other row of every block
    untouched.  fp32 and fp16 projections, fp32 ring."""
    import whisper_double as W
    from whisper_timestamped import streams
    W.install()
    model = W.build_model("tiny", seed=0, device=DEV)
    del model.alignment_heads                                    # -> the published tiny heads (parameter-count table)
    from whisper_timestamped.transcribe import get_alignment_heads
    heads = get_alignment_heads(model)
    hooked = list(range(len(model.decoder.blocks)))
    n_blocks, g = 6, torch.Generator().manual_seed(21)
    rings = streams.StreamRings(model, heads, hooked, n_blocks, torch.float32)
    D, H, hd, n_ctx = model.dims.n_text_state, model.dims.n_text_head, 64, 1500
    sl, sh, ss = (t.tolist() for t in rings.sel)
    for dtype, tol in ((torch.float32, 2e-5), (torch.float16, 2e-2)):
        rings.qk.fill_(-55.0)
        idx = [4, 0, 3]                                          # three of the six blocks take part in this call
        n_q = 5
        qs = [None if l not in rings.used else (torch.randn((len(idx), n_q, D), generator=g) * 0.7).to(dtype).to(DEV)
              for l in range(len(hooked))]
        ks = [None if l not in rings.used else (torch.randn((len(idx), n_ctx, D), generator=g) * 0.7).to(dtype).to(DEV)
              for l in range(len(hooked))]
        rings.write_qk(qs, ks, torch.tensor(idx, dtype=torch.int32, device=DEV), row=9)
        torch.cuda.synchronize()
        scale = hd ** -0.25
        for b, block in enumerate(idx):
            for i, h, slot in zip(sl, sh, ss):
                l = rings.used[i]
                want = (qs[l][b, -1].float() * scale)[h * hd:(h + 1) * hd] @ (ks[l][b].float() * scale)[:, h * hd:(h + 1) * hd].T
                got = rings.qk[block, slot, 9]
                assert (got - want).abs().max().item() <= tol, (dtype, block, slot)
        touched = torch.zeros_like(rings.qk, dtype=torch.bool)
        touched[idx, :, 9] = True
        assert (rings.qk[~touched] == -55.0).all()               # other rows, other blocks: as they were
    with pytest.raises(_lib().WtError, match="ring_index"):     # a null ring_index is refused before anything is launched
        rc = _lib().load().wt_qk_rows_streams(0, 0, 1, 0, 1, 1, 0, 0, 1500, 384, 64, 0.35, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0)
        _lib()._check(rc, "wt_qk_rows_streams")


def test_release_stream_frees_that_streams_arenas():
    """wt_release_stream: a side stream used for one call owns scratch arenas (the log-mel filterbank bands, per-unit words);
    releasing it frees them (count > 0), releasing it again frees nothing, and the stream can be used again afterwards."""
    L = _lib()
    g = torch.Generator(device=DEV).manual_seed(2)
    pcm = torch.randn((2, 48000), generator=g, device=DEV) * 0.1
    fb = O.mel_filters_ref(80)
    side = torch.cuda.Stream(device=DEV)
    with torch.cuda.stream(side):
        mel0, _ = L.logmel(pcm, fb, None, n_frames=300)
    side.synchronize()
    assert L.release_stream(side) >= 1
    assert L.release_stream(side) == 0
    with torch.cuda.stream(side):
        mel1, _ = L.logmel(pcm, fb, None, n_frames=300)
    side.synchronize()
    assert torch.equal(mel0, mel1)
    L.release_stream(side)


def test_cost_rejects_more_than_256_rows():
    L = _lib()
    descs = L.make_descs(1)
    descs[0]["T"], descs[0]["F"], descs[0]["pad_from"] = 257, 300, -1
    descs[0]["head_stride"], descs[0]["row_stride"] = 257 * 1500, 1500
    n_cost, _, _ = L.layout_outputs(descs)
    qk = torch.zeros((2, 257, 1500), device=DEV)
    with pytest.raises(L.WtError, match="unsupported shape"):
        L.cost_batch(qk, descs, L.descs_to_device(descs, DEV), torch.arange(2, dtype=torch.int32, device=DEV),
                     torch.empty(n_cost, device=DEV))


def test_logprob_gather_strided_rows_and_suppressed_token():
    L = _lib()
    rng = np.random.RandomState(5)
    big = torch.from_numpy(rng.standard_normal((6, 3, 5000)).astype(np.float32)).to(DEV)
    rows = big[:, 1, :]                                   # row stride 15000, unaligned starts
    toks = torch.tensor([0, 4999, 17, 3, 2500, 1], dtype=torch.int32)
    got = L.logprob_gather(rows, toks).cpu()
    want = O.token_logprob_gather_ref(rows.cpu(), toks.numpy())
    assert (got - want).abs().max() < 2e-5
    mask = torch.zeros(5000, dtype=torch.bool)
    mask[17] = True
    got = L.logprob_gather(rows, toks, mask).cpu()
    assert got[2] == -np.inf and torch.isfinite(got[[0, 1, 3, 4, 5]]).all()


def test_find_start_padding_fixture_and_random():
    L = _lib()
    mels, exp = [], []
    for p in json.load(open(os.path.join(G, "find_start_padding.json"))):
        if p["n_mels"] != 80:
            continue
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
        mels.append(m[0])
        exp.append(-1 if p["expected"] is None else p["expected"])
    rng = np.random.RandomState(6)
    for _ in range(20):
        m = rng.standard_normal((80, 3000)).astype(np.float32)
        c = int(rng.randint(0, 3001))
        m[:, c:] = 0.0
        if rng.rand() < 0.3:
            m[:, -1] = -0.0
        mels.append(m)
        r = O.find_start_padding_ref(torch.from_numpy(m[None]))
        exp.append(-1 if r is None else r)
    got = L.find_start_padding(torch.from_numpy(np.stack(mels)).to(DEV)).cpu().tolist()
    assert got == exp
    m128 = np.zeros((2, 128, 3000), np.float32)
    m128[0, 5, :1] = 1.0
    m128[1, 127, :2999] = 1.0
    assert L.find_start_padding(torch.from_numpy(m128).to(DEV)).cpu().tolist() == [0, 2999]


def test_logmel_vs_oracle():
    L = _lib()
    rng = np.random.RandomState(7)
    n = 480000
    pcm = (0.1 * rng.standard_normal((3, n))).astype(np.float32)
    t = np.arange(n) / 16000.0
    pcm[1] += (0.3 * np.sin(2 * np.pi * 440 * t)).astype(np.float32)
    valid = np.array([n, n, 16000 * 7 + 77], dtype=np.int32)
    pcm[2, valid[2]:] = 0.0
    for n_mels in (80, 128):
        fb = O.mel_filters_ref(n_mels)
        mel, gmax = L.logmel(torch.from_numpy(pcm).to(DEV), fb, torch.from_numpy(valid))
        mel = mel.cpu()
        for b in range(3):
            ref = O.pad_or_trim_ref(O.log_mel_spectrogram_ref(torch.from_numpy(pcm[b, :valid[b]]), n_mels), 3000)
            err = (mel[b] - ref).abs().max().item()
            assert err < 2e-4, (n_mels, b, err)
        # exact zeros in the padding, so that find_start_padding sees it
        assert (mel[2][:, valid[2] // 160:] == 0).all()
        assert L.find_start_padding(mel.to(DEV)).cpu().tolist()[2] == valid[2] // 160


def test_logmel_pad_batch_equals_the_detector_run_afterwards():
    """wt_logmel_pad_batch: find_start_padding of the windows it has just written (transcribe.py:1795-1805: the walk starts
    at the last valid column, the columns behind it being zeros by construction) == wt_find_start_padding_batch on the finished windows == the oracle's walk, for full windows
    (None), ragged ones, tile-boundary and nearly empty ones, a silent one, 80 and 128 mel bins, window sizes that take
    the 16-byte and the scalar path of the pass -- and the mel / max it returns are those of wt_logmel_batch."""
    L = _lib()
    rng = np.random.RandomState(23)
    n = 480000
    lengths = [n, 16000 * 7 + 77, 160 * 12 * 100, 201, n - 1, 160 * 2998, 160 * 2999, 160 * 3, n, 16000 * 29 + 159, 399, 160]
    pcm = (0.1 * rng.standard_normal((len(lengths), n))).astype(np.float32)
    for b, m in enumerate(lengths):
        pcm[b, m:] = 0.0
    pcm[8, :] = 0.0                                       # a full-length window of digital silence (no padding: None)
    valid = torch.tensor(lengths, dtype=torch.int32)
    for n_mels, n_frames in ((80, 3000), (128, 3000), (80, 2999), (80, 1501)):
        fb = O.mel_filters_ref(n_mels)
        mel0, gmax0 = L.logmel(torch.from_numpy(pcm).to(DEV), fb, valid, n_frames=n_frames)
        mel1, gmax1, pad = L.logmel(torch.from_numpy(pcm).to(DEV), fb, valid, n_frames=n_frames, with_padding=True)
        assert torch.equal(mel0, mel1) and torch.equal(gmax0, gmax1)
        want = L.find_start_padding(mel0).cpu().tolist()
        assert pad.cpu().tolist() == want, (n_mels, n_frames, pad.cpu().tolist(), want)
        for b in range(len(lengths)):
            r = O.find_start_padding_ref(mel0[b:b + 1].cpu())
            assert want[b] == (-1 if r is None else int(r)), (n_mels, n_frames, b)
        if n_frames == 3000:
            assert want[0] == -1 and want[8] == -1 and want[1] == lengths[1] // 160 and want[3] == 0 and want[11] == 0 and want[7] == 3
    # no valid samples argument at all: every window is full -> None everywhere; twice on the same stream (the per-chunk
    # word the votes merge into is re-zeroed by every call)
    for _ in range(2):
        _, _, pad = L.logmel(torch.from_numpy(pcm[:3]).to(DEV), O.mel_filters_ref(80), None, with_padding=True)
        assert pad.cpu().tolist() == [-1, -1, -1]


def test_logmel_batch_walks_tiles():
    """The STFT kernel is persistent: with more tiles than resident workgroups (4 per CU) a workgroup walks several
    tiles, prefetching the next span while it computes.  A ragged batch (full, short, tile-boundary and nearly empty
    chunks in every order, so that interior, reflecting and all-padding tiles follow each other) must give, chunk by
    chunk, exactly what the one-chunk call gives (one tile per workgroup, nothing prefetched across tiles) -- also with
    sample counts that are not a multiple of 4 and a base pointer that is not 16-byte aligned (scalar span loads)."""
    L = _lib()
    rng = np.random.RandomState(11)
    n = 480000
    lengths = [n, 160 * 12 * 7, 201, n - 1, 16000 * 11 + 3, 160 * 12 * 100 + 5, 399, n, 160 * 1500, 1000, n - 160 * 12, 16000 * 29]
    lengths = lengths + lengths[::-1]                              # 24 chunks x 250 tiles: ~6 tiles per workgroup
    pcm = (0.1 * rng.standard_normal((len(lengths), n))).astype(np.float32)
    for b, m in enumerate(lengths):
        pcm[b, m:] = 0.0
    valid = torch.tensor(lengths, dtype=torch.int32)
    fb = O.mel_filters_ref(80)
    dev_pcm = torch.from_numpy(pcm).to(DEV)
    mel, gmax = L.logmel(dev_pcm, fb, valid)
    for b, m in enumerate(lengths):
        one, g1 = L.logmel(dev_pcm[b:b + 1], fb, valid[b:b + 1])
        assert torch.equal(mel[b], one[0]), (b, m)
        assert torch.equal(gmax[b], g1[0]), (b, m)
    for b in (0, 2, 4, 5):                                         # and the oracle on a few of them
        ref = O.pad_or_trim_ref(O.log_mel_spectrogram_ref(torch.from_numpy(pcm[b, :lengths[b]]), 80), 3000)
        assert (mel[b].cpu() - ref).abs().max().item() < 2e-4, b
    # scalar span loads: rows of n - 2 samples carved out of a buffer one float off a 16-byte boundary
    flat = torch.zeros(len(lengths) * (n - 2) + 1, dtype=torch.float32, device=DEV)
    odd = flat[1:].view(len(lengths), n - 2)
    odd.copy_(dev_pcm[:, :n - 2])
    assert odd.data_ptr() % 16 == 4
    v2 = torch.clamp(valid, max=n - 2)
    mel2, _ = L.logmel(odd, fb, v2)
    for b in (0, 1, 3, 7):
        one, _ = L.logmel(dev_pcm[b:b + 1, :n - 2].contiguous(), fb, v2[b:b + 1])
        assert torch.equal(mel2[b], one[0]), b


def test_reference_side_stub():
    """Executes the ctypes stub printed in INTEGRATION.md section 3 (what a maintainer of the reference would
    add) against the oracle, so the documented binding cannot rot."""
    import ctypes
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md"), encoding="utf-8").read()
    code = re.search(r"```python\n# --- whisper_timestamped/transcribe.py \(reference side\).*?\n(.*?)```", text, re.S).group(1)
    code = code.replace('ctypes.CDLL("libwtalign.so")', f'ctypes.CDLL({_lib().LIB_PATH!r})')
    ns = {}
    exec(code, ns)
    L, H, T = 6, 8, 21
    heads = [(3, 1), (4, 2), (4, 3), (4, 7), (5, 1), (5, 2), (5, 4), (5, 6)]
    qk = synth.synth_qk(77, L * H, T, lo=100, hi=420).reshape(L, H, T, 1500)
    mask = torch.zeros(L, H, dtype=torch.bool)
    for l, h in heads:
        mask[l, h] = True
    start, end, max_duration = 90, 430, 200
    jumps, cost = ns["_wt_jumps"](torch.from_numpy(qk).to(DEV), start, end, mask.to_sparse(), max_duration)
    sel = O.select_heads_ref([torch.from_numpy(qk[l:l + 1]) for l in range(L)], start, end, np.array(heads))
    want_cost = O.cost_matrix_ref(sel, 9, 1.0, max_duration, start)
    r = O.dtw_ref(want_cost)
    want = O.jumps_from_path(r.index1s, r.index2s)
    got_cost = cost.cpu().numpy().reshape(T, end - start)
    np.testing.assert_allclose(got_cost, want_cost, rtol=2e-5, atol=2e-7)
    assert np.abs(jumps.astype(np.int64) - want).max() <= 1
    jumps2, _, starts = ns["_wt_jumps"](torch.from_numpy(qk).to(DEV), start, end, mask.to_sparse(), max_duration, True)
    assert np.array_equal(jumps2, jumps)
    assert np.array_equal(starts.astype(np.int64), O.jumps_start_ref(got_cost, jumps.astype(np.int64)))


def test_disfluency_kernel_vs_scipy():
    """wt_disfluency_batch (T.py:1656-1672: where the last attention peak of every token span starts) against the
    oracle, which calls scipy.signal.find_peaks the way the reference does.  Integer output: must be identical.
    Units of every shape class, profiles with several bumps per span, a third of them quantised (plateaus, ties)."""
    from test_oracle import _peaky_cost
    lib = _lib()
    L = lib.load()
    rng = np.random.RandomState(17)
    shapes = [(1, 40), (3, 400), (7, 1500), (12, 1792), (30, 1500), (64, 1500), (130, 1500), (224, 1500), (256, 1792), (2, 3)]
    descs = lib.make_descs(len(shapes))
    costs, jumps = [], []
    for k, (d, (T, F)) in enumerate(zip(descs, shapes)):
        d["T"], d["F"] = T, F
        costs.append(_peaky_cost(rng, T, F, quantised=k % 3 == 0, density=max(1, F // 100)))
        cuts = np.sort(rng.randint(0, F, size=T - 1)) if T > 1 else np.zeros(0, dtype=np.int64)
        jumps.append(np.concatenate([[0], cuts, [F - 1]]).astype(np.int32))
    n_cost, n_jumps, _ = lib.layout_outputs(descs)
    cost = torch.zeros(n_cost, dtype=torch.float32)
    jp = torch.zeros(n_jumps, dtype=torch.int32)
    for d, c, j in zip(descs, costs, jumps):
        cost[int(d["cost_offset"]):int(d["cost_offset"]) + c.size] = torch.from_numpy(c.ravel())
        jp[int(d["jumps_offset"]):int(d["jumps_offset"]) + j.size] = torch.from_numpy(j)
    cost, jp = cost.to(DEV), jp.to(DEV)
    out = torch.full_like(jp, -7)
    dd = lib.descs_to_device(descs, DEV)
    lib._check(L.wt_disfluency_batch(cost.data_ptr(), dd.data_ptr(), len(shapes), jp.data_ptr(), out.data_ptr(), 0.02, 3.0,
                                      lib._stream()), "wt_disfluency_batch")
    out = out.cpu().numpy()
    moved = 0
    for d, c, j in zip(descs, costs, jumps):
        want = O.jumps_start_ref(c, j.astype(np.int64))
        got = out[int(d["jumps_offset"]):int(d["jumps_offset"]) + j.size]
        assert np.array_equal(got, want), (int(d["T"]), int(d["F"]), np.nonzero(got != want)[0][:5])
        moved += int((want != j).sum())
    assert moved > 20
    assert L.wt_disfluency_batch(0, dd.data_ptr(), 1, jp.data_ptr(), out.ctypes.data, 0.02, 3.0, 0) == -1     # WT_E_BADARG


def test_per_device_state_second_device_first():
    """The library keeps per-DEVICE state (the log-mel __constant__ tables, the scratch arenas, the DTW kernels' dynamic-LDS
    attribute): a process that touches cuda:1 BEFORE cuda:0 must get the same results on both (round 2 uploaded the
    tables once per process: the second device computed log-mel from zeros).  Needs two visible GPUs."""
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: the per-device paths cannot be exercised on this box")
    L = _lib()
    from whisper_timestamped.audio import mel_filters
    g = torch.Generator().manual_seed(3)
    pcm = torch.randn((2, 48000), generator=g) * 0.1
    want = torch.stack([O.pad_or_trim_ref(O.log_mel_spectrogram_ref(pcm[b], 80), 3000) for b in range(2)])
    costs = [np.random.RandomState(5).standard_normal((70, 300)).astype(np.float32) - 3.0]
    ref = O.dtw_ref(costs[0].astype(np.float64))
    for dev in ("cuda:1", "cuda:0"):
        mel, _ = L.logmel(pcm.to(dev), mel_filters(dev, 80), None, 3000)
        assert (mel.cpu() - want).abs().max() < 2e-4, dev
        with torch.cuda.device(dev):
            descs = L.make_descs(1)
            descs[0]["T"], descs[0]["F"], descs[0]["pad_from"] = 70, 300, -1
            n_cost, n_jumps, _ = L.layout_outputs(descs)
            flat = np.zeros(n_cost, dtype=np.float32)
            flat[:70 * 300] = costs[0].ravel()
            jumps = torch.zeros(n_jumps, dtype=torch.int32, device=dev)
            L.dtw_batch(torch.from_numpy(flat).to(dev), descs, L.descs_to_device(descs, dev), jumps)
            torch.cuda.synchronize(dev)
        assert np.array_equal(jumps.cpu().numpy(), O.jumps_from_path(ref.index1s, ref.index2s)), dev


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_cost_and_jumps_do_not_depend_on_the_batch(dtype):
    """A unit's cost matrix and jumps are a function of the unit alone: the same unit aligned alone, in a batch of its own
    F class, and in a batch that holds every F class (short units then share a launch with the (256, 512] class) gives
    bit-identical results.  (Round 2 served short units with the 8-elements-per-lane rowmean whenever the batch also held
    a (256, 512] unit: last-place differences in the softmax denominator.)"""
    L = _lib()
    probe = [(11, 3, 147), (64, 0, 256), (9, 100, 245), (30, 40, 392), (17, 275, 523), (70, 100, 400), (12, 0, 700), (6, 0, 1100)]
    others = [(20, 0, 300), (33, 10, 500), (40, 0, 600), (12, 0, 900), (9, 0, 1300), (100, 0, 1500), (5, 1, 8), (64, 200, 456)]
    alone = []
    for k, sh in enumerate(probe):
        r = _align_units([sh], dtype, L.WT_ALIGN_KEEP_COST, seed=500 + k, want_path=False)
        d = r["descs"][0]
        T, F = int(d["T"]), int(d["F"])
        alone.append((r["cost"][:T * F].clone(), r["jumps"][:T + 1].clone()))

    def in_batch(shapes, seeds):
        # (_align_units seeds unit k with seed + k: build the batch unit by unit so that every probe keeps its own data)
        qk_list = [synth.synth_qk(sd, 8, T, lo=s, hi=e) for sd, (T, s, e) in zip(seeds, shapes)]
        order = L.launch_order([(T, e - s) for T, s, e in shapes])
        descs = L.make_descs(len(shapes))
        offs, off = [], 0
        for q in qk_list:
            offs.append(off)
            off += q.size
        for d, i in zip(descs, order):
            T, s, e = shapes[i]
            q = qk_list[i]
            d["qk_offset"], d["head_stride"], d["row_stride"] = offs[i], q.shape[1] * q.shape[2], q.shape[2]
            d["T"], d["F"], d["start_token"], d["pad_from"] = T, e - s, s, (-1 if i % 3 else max((e - s) // 2, 1))
        n_cost, n_jumps, _ = L.layout_outputs(descs)
        qk = torch.from_numpy(np.concatenate([q.ravel() for q in qk_list])).to(DEV).to(dtype)
        cost = torch.zeros(n_cost, dtype=torch.float32, device=DEV)
        jumps = torch.zeros(n_jumps, dtype=torch.int32, device=DEV)
        L.align_batch(qk, descs, L.descs_to_device(descs, DEV), torch.arange(8, dtype=torch.int32, device=DEV), cost, jumps)
        torch.cuda.synchronize()
        return {i: (d, cost, jumps) for d, i in zip(descs, order)}

    # unit i of _align_units([sh]) had pad index rule "i % 3" with i = 0 -> a mask: give the probes index 0 mod 3 in the batch
    shapes, seeds, where = [], [], []
    for k, sh in enumerate(probe):
        where.append(len(shapes))
        shapes.append(sh)
        seeds.append(500 + k)
        for j in range(2):                      # two fillers keep the next probe at an index divisible by 3
            shapes.append(others[(2 * k + j) % len(others)])
            seeds.append(900 + 2 * k + j)
    got = in_batch(shapes, seeds)
    for (c_alone, j_alone), w in zip(alone, where):
        d, cost, jumps = got[w]
        T, F, c0, j0 = int(d["T"]), int(d["F"]), int(d["cost_offset"]), int(d["jumps_offset"])
        assert torch.equal(cost[c0:c0 + T * F], c_alone), f"cost of unit (T={T}, F={F}) depends on the batch"
        assert torch.equal(jumps[j0:j0 + T + 1], j_alone), (T, F)


def test_fused_small_units_batch_larger_than_the_chip():
    """More small units than the chip has CUs: the fused tail kernel is launched per (single-/multi-wave, light/heavy LDS)
    class instead of once, and the row pass takes its per-class launches (more than 4096 token rows): still bit-identical
    to the batched kernels, unit by unit."""
    L = _lib()
    rng = np.random.RandomState(77)
    shapes = [(64, 0, 250), (74, 100, 400), (130, 0, 60), (30, 0, 1025), (16, 0, 1500)]      # heavy / multi-wave members
    Ts, Fs = synth.draw_real_shapes(123, 330)
    for T, F in zip(Ts, Fs):
        T, F = int(min(T, 60)), int(min(F, 700))
        F = max(F, T + 1)
        s = int(rng.randint(0, 1500 - F + 1))
        shapes.append((T, s, s + F))
    assert sum(L.small_unit(T, e - s) for T, s, e in shapes) > 256
    fused = _align_units(shapes, torch.float32, L.WT_ALIGN_KEEP_COST, seed=9000, want_path=False)
    plain = _align_units(shapes, torch.float32, L.WT_ALIGN_KEEP_COST | L.WT_ALIGN_NO_FUSED_SMALL_UNITS, seed=9000, want_path=False)
    assert torch.equal(fused["jumps"], plain["jumps"])
    for d in fused["descs"]:
        c0, n = int(d["cost_offset"]), int(d["T"]) * int(d["F"])
        assert torch.equal(fused["cost"][c0:c0 + n], plain["cost"][c0:c0 + n]), (int(d["T"]), int(d["F"]))
    d = fused["descs"][0]
    T, F = int(d["T"]), int(d["F"])
    r = O.dtw_ref(fused["cost"][:T * F].reshape(T, F).cpu().numpy().astype(np.float64))
    assert np.array_equal(fused["jumps"][:T + 1].cpu().numpy(), O.jumps_from_path(r.index1s, r.index2s))
