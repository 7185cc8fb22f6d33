#!/usr/bin/env python3
"""Generate tests/golden/* by running THE REFERENCE'S OWN CODE in the build
container (it cannot travel to the GPU box, the fixtures can).

/root/reference/whisper_timestamped/transcribe.py is loaded as a stand-alone
module with stub modules for the third-party packages that are absent here:

  * ``whisper``  -- only what transcribe.py touches at import time
                    (``__version__``, ``utils.format_timestamp``, ``audio``
                    constants, ``model.disable_sdpa``);
  * ``dtw``      -- ``dtw.dtw`` / ``dtw.stepPattern`` backed by the oracle's generic
                    step-pattern interpreter (oracle/dtw_patterns.py): the reference's own
                    ``StepPattern(_c(...))`` expression (transcribe.py:1575-1580) is executed,
                    its rows interpreted; every call is cross-checked against the C
                    restatement (oracle/dtw_ref.c).  The stub also RECORDS the local-cost
                    matrix the reference hands to it, which is how the f64 cost built by
                    transcribe.py:1540-1568 is captured.

Everything else (scipy.ndimage.median_filter, torch CPU ops, jumps, word
boundaries, rounding, find_start_padding, the post-fixers) is the reference's
code executing unmodified.

Run:  python tests/golden/make_golden.py      (needs /root/reference)
"""
import importlib.util
import json
import os
import sys
import types
from contextlib import contextmanager

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import align_ref as O  # noqa: E402
import synth  # noqa: E402

REF = "/root/reference/whisper_timestamped/transcribe.py"

_captured = {}


def load_reference():
    w = types.ModuleType("whisper")
    w.__version__ = "20240930"
    wu = types.ModuleType("whisper.utils")
    wu.format_timestamp = lambda *a, **k: ""
    wa = types.ModuleType("whisper.audio")
    wa.N_FRAMES, wa.HOP_LENGTH, wa.SAMPLE_RATE = 3000, 160, 16000
    wm = types.ModuleType("whisper.model")

    @contextmanager
    def disable_sdpa():
        yield

    wm.disable_sdpa = disable_sdpa
    wm.TextDecoder = type("TextDecoder", (torch.nn.Module,), {})   # only subclassed at import (transcribe.py:2925,2953)
    wm.Whisper = type("Whisper", (torch.nn.Module,), {})
    w.utils, w.audio, w.model = wu, wa, wm
    sys.modules.update({"whisper": w, "whisper.utils": wu, "whisper.audio": wa, "whisper.model": wm})

    from oracle import dtw_patterns as P

    def record(x, res):
        _captured["cost"] = np.array(x, dtype=np.float64, copy=True)
        _captured["index1s"], _captured["index2s"] = res.index1s, res.index2s

    def c_restatement(x, pattern):          # the hard-coded form: 3 moves = symmetric1, 2 moves = the reference's own pattern
        r = O.dtw_ref(x, step_pattern=0 if pattern.n_patterns == 3 else 1)
        return r.index1s, r.index2s
    sys.modules.update(P.stub_modules(on_call=record, cross_check=c_restatement))

    spec = importlib.util.spec_from_file_location("ref_transcribe", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.num_alignment_for_plot = 0   # module global initialised by transcribe_timestamped (transcribe.py:300-301)
    return mod


def heads_sparse(pairs, L, H):
    m = torch.zeros(L, H, dtype=torch.bool)
    for l, h in pairs:
        m[l, h] = True
    return m.to_sparse()


BASE_HEADS = [(3, 1), (4, 2), (4, 3), (4, 7), (5, 1), (5, 2), (5, 4), (5, 6)]
TINY_HEADS = [(2, 2), (3, 0), (3, 2), (3, 3), (3, 4), (3, 5)]


def align_case_list():
    """Parameter sets; inputs are regenerated from these by tests (synth.py)."""
    C = []
    # (name, seed, L, H, heads, n_text, start_frame, end_frame, refine, pad_col, disfl, use_space, multilingual)
    C.append(dict(name="tiny_short", seed=101, L=4, H=6, heads=TINY_HEADS, n_text=6, start=0, end=120, refine=25))
    C.append(dict(name="tiny_allheads", seed=102, L=4, H=6, heads=None, n_text=9, start=40, end=260, refine=25))
    C.append(dict(name="base_typical", seed=103, L=6, H=8, heads=BASE_HEADS, n_text=16, start=300, end=498, refine=25))
    C.append(dict(name="base_norefine", seed=104, L=6, H=8, heads=BASE_HEADS, n_text=12, start=100, end=330, refine=0))
    C.append(dict(name="base_pad", seed=105, L=6, H=8, heads=BASE_HEADS, n_text=20, start=200, end=700, refine=25,
                  pad_col=900))     # mfcc zero from mel column 900 -> max_duration 450 (< start 175? no: >) masks [:, 450:]
    C.append(dict(name="base_pad_outside", seed=106, L=6, H=8, heads=BASE_HEADS, n_text=8, start=600, end=800,
                  refine=25, pad_col=1000))  # start_token 575 >= max_duration 500 -> warning only
    C.append(dict(name="base_disfl", seed=107, L=6, H=8, heads=BASE_HEADS, n_text=14, start=0, end=420, refine=25,
                  disfl=True))
    C.append(dict(name="base_edge_end", seed=108, L=6, H=8, heads=BASE_HEADS, n_text=10, start=1380, end=1500,
                  refine=25))       # window clamps at 1500
    C.append(dict(name="base_toomuchtext", seed=109, L=6, H=8, heads=BASE_HEADS, n_text=40, start=700, end=710,
                  refine=0))        # T > F -> min-duration rule then exact fit
    C.append(dict(name="base_truncate", seed=110, L=6, H=8, heads=BASE_HEADS, n_text=30, start=1480, end=1500,
                  refine=0))        # T=32 > F=20 -> recursion :1516-1535, unfinished_decoding
    C.append(dict(name="base_unicode", seed=111, L=6, H=8, heads=BASE_HEADS, n_text=10, start=50, end=300, refine=25,
                  use_space=False))
    C.append(dict(name="base_en", seed=112, L=6, H=8, heads=[(3, 3), (4, 7), (5, 1), (5, 5), (5, 7)], n_text=11,
                  start=10, end=200, refine=25, multilingual=False))
    C.append(dict(name="base_noend", seed=113, L=6, H=8, heads=BASE_HEADS, n_text=15, start=900, end=None, refine=25))
    C.append(dict(name="base_long", seed=114, L=6, H=8, heads=BASE_HEADS, n_text=58, start=0, end=1500, refine=25))
    C.append(dict(name="tiny_F_small", seed=115, L=4, H=6, heads=TINY_HEADS, n_text=1, start=10, end=12, refine=0))
    C.append(dict(name="base_rmpunct", seed=116, L=6, H=8, heads=BASE_HEADS, n_text=18, start=20, end=380, refine=25,
                  remove_punct=True))
    # subwords_can_be_empty=False: the reference's own second step pattern (transcribe.py:1575-1580)
    C.append(dict(name="base_noempty", seed=117, L=6, H=8, heads=BASE_HEADS, n_text=15, start=120, end=420, refine=25,
                  noempty=True))
    C.append(dict(name="tiny_noempty_tight", seed=118, L=4, H=6, heads=TINY_HEADS, n_text=22, start=700, end=726, refine=0,
                  noempty=True))    # 24 tokens on 26 frames: almost every token gets exactly one frame
    return C


def build_case_inputs(c):
    """Shared with tests: params -> (tokens, attention list, heads pairs, mfcc, tokenizer)."""
    tok = synth.StubTokenizer(multilingual=c.get("multilingual", True))
    end = c["end"]
    tokens = synth.synth_segment_tokens(c["seed"], c["n_text"], c["start"], end if end is not None else 0, tok,
                                        with_end=end is not None)
    if end is None:
        tokens.append(tok.eot)
    T = len(tokens)
    lo = max(c["start"] - 25, 0)
    hi = min((end if end is not None else 1500) + 25, 1500)
    qk = synth.synth_qk(c["seed"], c["L"] * c["H"], T, lo=lo, hi=max(hi, lo + 1))
    qk = qk.reshape(c["L"], 1, c["H"], T, 1500)
    att = [torch.from_numpy(qk[l]) for l in range(c["L"])]
    mfcc = None
    if c.get("pad_col") is not None:
        rng = np.random.RandomState(c["seed"] + 1)
        m = rng.standard_normal((1, 80, 3000)).astype(np.float32)
        m[..., c["pad_col"]:] = 0.0
        mfcc = torch.from_numpy(m)
    return tokens, att, c["heads"], mfcc, tok


def main():
    ref = load_reference()
    out_cases = []
    costs = {}
    for c in align_case_list():
        tokens, att, heads, mfcc, tok = build_case_inputs(c)
        ah = None if heads is None else heads_sparse(heads, c["L"], c["H"])
        _captured.clear()
        words = ref.perform_word_alignment(
            list(tokens), att, tok,
            use_space=c.get("use_space", True),
            mfcc=mfcc,
            refine_whisper_precision_nframes=c["refine"],
            remove_punctuation_from_words=c.get("remove_punct", False),
            alignment_heads=ah,
            detect_disfluencies=c.get("disfl", False),
            subwords_can_be_empty=not c.get("noempty", False),
        )
        rec = dict(c)
        rec["words"] = [dict(text=w["text"], start=w["start"], end=w["end"], tokens=w["tokens"],
                             tokens_indices=[int(x) for x in w["tokens_indices"]]) for w in words]
        rec["index1s"] = _captured["index1s"].tolist()
        rec["index2s"] = _captured["index2s"].tolist()
        cost = _captured["cost"]
        assert np.array_equal(cost.astype(np.float32).astype(np.float64), cost), "cost not f32-exact"
        costs[c["name"]] = cost.astype(np.float32)
        rec["cost_shape"] = list(cost.shape)
        out_cases.append(rec)
        print(f"{c['name']:18s} T,F={cost.shape} words={len(words)}"
              f" first={rec['words'][0]['text'] if words else None!r}")

    with open(os.path.join(HERE, "align_cases.json"), "w", encoding="utf-8") as f:
        json.dump(out_cases, f, ensure_ascii=False, indent=0)
    np.savez_compressed(os.path.join(HERE, "align_cost.npz"), **costs)

    # ---- find_start_padding (transcribe.py:1795-1805) -------------------
    pads = []
    for seed, n_mels, col, kind in [(1, 80, 900, "zeros"), (2, 80, None, "none"), (3, 128, 1, "zeros"),
                                    (4, 80, 0, "allzero"), (5, 80, 2999, "zeros"), (6, 80, 1500, "const_nonzero"),
                                    (7, 80, 2000, "zero_col_inside")]:
        rng = np.random.RandomState(seed)
        m = rng.standard_normal((1, n_mels, 3000)).astype(np.float32)
        if kind in ("zeros", "zero_col_inside"):
            m[..., col:] = 0.0
            if kind == "zero_col_inside":
                m[..., 1000] = 0.0  # an all-zero column inside the signal must not matter
        elif kind == "allzero":
            m[:] = 0.0
        elif kind == "const_nonzero":
            m[..., col:] = 0.5
        r = ref.find_start_padding(torch.from_numpy(m))
        pads.append(dict(seed=seed, n_mels=n_mels, col=col, kind=kind, expected=r))
    with open(os.path.join(HERE, "find_start_padding.json"), "w") as f:
        json.dump(pads, f, indent=0)

    # ---- word split KAT replay through the reference code with the stub tokenizer
    kat = json.load(open(os.path.join(HERE, "split_tokens_kat.json"), encoding="utf-8"))
    for k in kat:
        tok = synth.StubTokenizer(multilingual=k["multilingual"])
        got = ref.split_tokens_on_spaces(list(k["tokens"]), tok)
        exp = (k["words"], k["word_tokens"], k["word_tokens_indices"])
        assert got == tuple(exp), (got, exp)
    print("reference split_tokens_on_spaces replays the KAT with StubTokenizer: OK")

    # ---- post-fixers (transcribe.py:2202-2295) ---------------------------
    import copy
    fix_cases = []
    rng = np.random.RandomState(5)
    for k in range(12):
        n = int(rng.randint(1, 12))
        t = np.round(np.abs(np.sort(rng.uniform(0, 20, size=2 * n)) + rng.normal(0, 0.3, size=2 * n)), 2)
        if k % 3 == 0:
            t = np.round(np.sort(t), 2)
        words = [dict(text=f"w{i}", start=float(t[2 * i]), end=float(t[2 * i + 1])) for i in range(n)]
        if k % 4 == 1:
            words[-1]["end"] = words[-1]["start"]
        seg = dict(start=float(min(t)), end=float(max(t)), words=words)
        inp = copy.deepcopy(seg)
        outp = copy.deepcopy(seg)
        try:
            ref.ensure_increasing_positions(outp["words"], min_duration=0.02 if k % 2 else 0)
            expected = outp["words"]
        except AssertionError:
            expected = "AssertionError"
        fix_cases.append(dict(kind="ensure_increasing_positions", min_duration=0.02 if k % 2 else 0,
                              input=inp["words"], expected=expected))
    with open(os.path.join(HERE, "postfix_cases.json"), "w") as f:
        json.dump(fix_cases, f, indent=0)
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
