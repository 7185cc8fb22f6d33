"""CPU tests: the C-ABI library loads and exports every symbol of
include/wtalign.h; the host-side (Python) part of the path reproduces the
reference's known answers and the fixtures generated from the reference."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

from oracle import align_ref as O
import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def test_library_exports_every_declared_symbol():
    from whisper_timestamped import _lib
    hdr = open(os.path.join(ROOT, "include", "wtalign.h")).read()
    declared = sorted(set(re.findall(r"\b(wt_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == sorted(_lib.EXPORTS)
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    L.wt_version.restype = ctypes.c_int
    assert L.wt_version() == _lib.ABI_VERSION == 5


def test_dynamic_symbol_table_is_exactly_the_header():
    """-fvisibility=hidden + csrc/wtalign.map: `nm -D` shows the WT_API entries and nothing else of ours (no mangled
    C++ internals, no kernel handle objects)."""
    import subprocess
    from whisper_timestamped import _lib
    hdr = open(os.path.join(ROOT, "include", "wtalign.h")).read()
    declared = sorted(set(re.findall(r"WT_API[^;(]*?\b(wt_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == sorted(_lib.EXPORTS)
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    defined = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert defined == declared, sorted(set(defined) ^ set(declared))


def test_release_stream_of_an_unknown_stream_frees_nothing():
    from whisper_timestamped import _lib
    assert _lib.release_stream(0x1234) == 0


def test_seg_desc_layout_matches_header():
    from whisper_timestamped import _lib
    assert _lib.SEG_DTYPE.itemsize == 64
    assert [_lib.SEG_DTYPE.fields[n][1] for n in ("qk_offset", "head_stride", "row_stride", "cost_offset",
                                                    "jumps_offset", "path_offset", "T", "F", "start_token",
                                                    "pad_from")] == [0, 8, 16, 24, 32, 40, 48, 52, 56, 60]


def test_product_refuses_cpu_tensors():
    import torch
    from whisper_timestamped import _lib
    with pytest.raises(_lib.WtError):
        _lib.find_start_padding(torch.zeros(1, 80, 3000))


def test_split_tokens_kat():
    from whisper_timestamped.words import split_tokens_on_spaces
    for k in json.load(open(os.path.join(G, "split_tokens_kat.json"), encoding="utf-8")):
        tok = synth.StubTokenizer(multilingual=k["multilingual"])
        got = split_tokens_on_spaces(list(k["tokens"]), tok)
        assert got == (k["words"], k["word_tokens"], k["word_tokens_indices"]), k["source"]


def test_split_tokens_random_vs_oracle():
    from whisper_timestamped.words import split_tokens_on_spaces, split_tokens_on_unicode
    tok = synth.StubTokenizer()
    rng = np.random.RandomState(11)
    for _ in range(300):
        n = rng.randint(1, 40)
        toks = [tok.timestamp_begin + int(rng.randint(0, 700))] + rng.randint(0, 50257, size=n).tolist()
        if rng.rand() < 0.8:
            toks.append(tok.timestamp_begin + int(rng.randint(700, 1500)))
        for rm in (False, True):
            try:
                want = O.split_tokens_on_spaces_ref(list(toks), tok, rm)
            except IndexError:
                with pytest.raises(IndexError):
                    split_tokens_on_spaces(list(toks), tok, rm)
                continue
            assert split_tokens_on_spaces(list(toks), tok, rm) == want
            assert split_tokens_on_unicode(list(toks), tok, rm) == O.split_tokens_on_unicode_ref(list(toks), tok, rm)


def test_frame_window_vs_oracle():
    from whisper_timestamped.words import frame_window
    rng = np.random.RandomState(12)
    tb = 50364
    for _ in range(2000):
        n = rng.randint(2, 60)
        toks = [tb + int(rng.randint(-2, 1501))] + rng.randint(0, 50000, size=n - 2).tolist() + \
               [int(rng.choice([tb + int(rng.randint(0, 1501)), 50257, tb + int(rng.randint(0, 40))]))]
        refine = int(rng.choice([0, 25]))
        try:
            want = O.frame_window_ref(toks, tb, refine)
        except RuntimeError:
            with pytest.raises(RuntimeError):
                frame_window(toks, tb, refine)
            continue
        assert frame_window(toks, tb, refine) == want


def test_words_from_fixture_jumps():
    """Host half of the path on the reference-generated fixtures: feed the
    fixture's DTW path, expect the fixture's words."""
    from whisper_timestamped import words as W
    from golden.make_golden import build_case_inputs
    cases = json.load(open(os.path.join(G, "align_cases.json"), encoding="utf-8"))
    for c in cases:
        if c.get("disfl"):
            continue
        tokens, att, heads, mfcc, tok = build_case_inputs(c)
        T, F = c["cost_shape"]
        unfinished = False
        if len(tokens) > T:              # the reference truncated (transcribe.py:1516-1535)
            tokens = tokens[:T - 1] + [tokens[-1]]
            unfinished = True
        win = W.frame_window(tokens, tok.timestamp_begin, c["refine"])
        assert win[1] - win[0] == F
        split = W.split_tokens_on_spaces if c.get("use_space", True) else W.split_tokens_on_unicode
        words, pieces, ids = split(tokens, tok, remove_punctuation_from_words=c.get("remove_punct", False))
        jumps = O.jumps_from_path(np.array(c["index1s"]), np.array(c["index2s"]))
        got = W.words_from_jumps(jumps, jumps, words, pieces, ids, W.trailing_punctuation_counts(pieces),
                                 win[0] * W.AUDIO_TIME_PER_TOKEN, c["refine"], unfinished)
        got = [dict(text=w["text"], start=w["start"], end=w["end"], tokens=w["tokens"],
                    tokens_indices=[int(x) for x in w["tokens_indices"]]) for w in got]
        assert got == c["words"], c["name"]


def test_ensure_increasing_positions_vs_reference_fixture():
    """postfix_cases.json: outputs of the reference's ensure_increasing_positions (transcribe.py:2265-2295)."""
    import copy
    from whisper_timestamped.postprocess import ensure_increasing_positions
    cases = json.load(open(os.path.join(G, "postfix_cases.json")))
    assert cases
    for c in cases:
        words = copy.deepcopy(c["input"])
        if c["expected"] == "AssertionError":
            with pytest.raises(AssertionError):
                ensure_increasing_positions(words, min_duration=c["min_duration"])
        else:
            assert ensure_increasing_positions(words, min_duration=c["min_duration"]) == c["expected"]


def test_alignment_head_tables_match_reference_dumps():
    """ALIGNMENT_HEADS equals the packed masks the reference carries (transcribe.py:2343-2357); SURVEY.md 8 table."""
    from whisper_timestamped.transcribe import ALIGNMENT_HEADS, alignment_heads_for
    assert ALIGNMENT_HEADS["base"][2] == [(3, 1), (4, 2), (4, 3), (4, 7), (5, 1), (5, 2), (5, 4), (5, 6)]
    assert len(ALIGNMENT_HEADS["small.en"][2]) == 19 and len(ALIGNMENT_HEADS["large-v2"][2]) == 23
    ah = alignment_heads_for("large-v3", 32, 20)
    assert ah.is_sparse and ah.coalesce().indices().T.tolist()[0] == [7, 0]


def test_package_surface_like_the_reference():
    """whisper_timestamped/__init__.py of the reference: transcribe / transcribe_timestamped / load_model /
    __version__ plus re-exported openai-whisper names (its KAT calls whisper_timestamped.tokenizer.get_tokenizer)."""
    import whisper_double
    whisper_double.install()
    import whisper_timestamped as wt
    assert wt.transcribe is wt.transcribe_timestamped and callable(wt.load_model) and wt.__version__.startswith("1.15.9")
    tk = wt.tokenizer.get_tokenizer(True, language=None)
    assert tk.timestamp_begin == 50364
    assert wt.DecodingOptions is whisper_double.DecodingOptions and wt.Whisper is whisper_double.Whisper
    for name in ("load_audio", "log_mel_spectrogram", "pad_or_trim", "available_models", "decode", "detect_language"):
        assert callable(getattr(wt, name))
    with pytest.raises(AttributeError):
        wt.no_such_name


def test_gpu_log_mel_patch_is_scoped(monkeypatch):
    """backend.gpu_log_mel instruments the backend's transcribe module only inside the context (also on errors)."""
    import sys
    import whisper_double
    whisper_double.install()
    from whisper_timestamped import backend
    mod = sys.modules["whisper.transcribe"]
    original = mod.log_mel_spectrogram
    with backend.gpu_log_mel("cuda:0", enabled=True) as on:      # no GPU here: the front-end self-check cannot pass,
        assert on is False and mod.log_mel_spectrogram is original   # so the backend keeps its own log-mel
    monkeypatch.setattr(backend, "_front_end_matches", lambda *a, **k: True)
    with backend.gpu_log_mel("cuda:0", enabled=False) as on:
        assert on is False and mod.log_mel_spectrogram is original
    with pytest.raises(RuntimeError):
        with backend.gpu_log_mel("cuda:0", enabled=True) as on:
            assert on is True and mod.log_mel_spectrogram is not original
            raise RuntimeError("decode failed")
    assert mod.log_mel_spectrogram is original


def test_c_abi_argument_validation_without_a_gpu():
    """Error behaviour of the boundary (include/wtalign.h): bad arguments are refused on the host, before any HIP
    call, with a negative code and a message -- checked here without a GPU."""
    from whisper_timestamped import _lib
    L = _lib.load()
    descs = _lib.make_descs(1)
    descs[0]["T"], descs[0]["F"] = 10, 100
    p = descs.ctypes.data
    assert L.wt_cost_batch(0, 0, p, p, 1, 0, 8, 9, 1.0, 0, 0) == -1 and b"null pointer" in L.wt_last_error()
    assert L.wt_cost_batch(p, 0, p, p, 1, p, 8, 8, 1.0, p, 0) == -3 and b"medfilt_width=8" in L.wt_last_error()      # even
    assert L.wt_cost_batch(p, 0, p, p, 1, p, 8, 11, 1.0, p, 0) == -3 and b"medfilt_width=11" in L.wt_last_error()    # > 9
    assert L.wt_cost_batch(p, 5, p, p, 0, p, 8, 9, 1.0, p, 0) == 0                     # n_seg == 0: nothing to do
    descs[0]["F"] = 5000
    assert L.wt_cost_batch(p, 0, p, p, 1, p, 8, 9, 1.0, p, 0) == -3 and b"unsupported shape" in L.wt_last_error()
    assert L.wt_dtw_batch(p, p, p, 1, p, 0, 0, 0, 0, 0) == -3 and b"unsupported shape" in L.wt_last_error()
    assert L.wt_dtw_batch(p, p, p, 1, p, p, 0, 0, 0, 0) == -1                          # path_i without path_j
    descs[0]["T"], descs[0]["F"] = 10, 100
    assert L.wt_dtw_batch_pattern(p, p, p, 1, 7, p, 0, 0, 0, 0, 0) == -1 and b"step_pattern=7" in L.wt_last_error()
    descs[0]["T"], descs[0]["F"] = 30, 20                                              # more tokens than frames
    assert L.wt_dtw_batch_pattern(p, p, p, 1, 1, p, 0, 0, 0, 0, 0) == -3 and b"no warping path" in L.wt_last_error()
    descs[0]["F"] = 5000
    assert L.wt_logprob_gather_batch(p, 0, 10, 4, 100, p, p, 0, p, 0) == -1            # row_stride < V
    assert L.wt_logprob_gather_batch(p, 0, 100, 4, 100, p, p, 2, p, 0) == -1           # suppress_rows not in {0,1,n}
    assert L.wt_logprob_gather_batch(p, 0, 100, 0, 100, p, 0, 0, p, 0) == 0
    assert L.wt_capture_rows(p, 0, 6, 1, 1500, p, p, 2, p, 0, 16, 16, 0) == -1 and b"row=16 of 16" in L.wt_last_error()
    assert L.wt_find_start_padding_batch(p, 1, 80, 1, p, 0) == -1
    assert L.wt_disfluency_batch(p, p, 1, 0, p, 0.02, 3.0, 0) == -1 and b"wt_disfluency_batch" in L.wt_last_error()
    assert L.wt_disfluency_batch(p, p, 0, p, p, 0.02, 3.0, 0) == 0                     # no units: nothing to do
    assert L.wt_logmel_batch(p, 1, 100, 0, p, 80, 3000, p, 0, 0) == -1                 # fewer than 201 samples
    descs[0]["T"], descs[0]["F"] = 300, 1500                                           # colnorm covers 256 token rows
    assert L.wt_cost_batch(p, 0, p, p, 1, p, 8, 9, 1.0, p, 0) == -3 and b"T=300" in L.wt_last_error()
    assert L.wt_logprob_gather_rows(p, 0, 100, 0, 4, 100, p, p, 0) == -1 and b"row_index" in L.wt_last_error()
    assert L.wt_logprob_gather_rows(p, 0, 100, p, 0, 100, p, p, 0) == 0                # no rows: nothing to do
    import ctypes as C
    ptrs = (C.c_void_p * 2)(p, p)
    args = lambda n_layers, hd, row0: (ptrs, ptrs, n_layers, 0, 2, 10, 5120, 768000, 1500, 512, hd, 0.35, p, p, p, 8, 0, 0, p,
                                       0, 8 * 16 * 1500, 16, row0, 0)   # noqa: E731
    assert L.wt_qk_rows_batch(*args(40, 64, 0)) == -1 and b"40 layers" in L.wt_last_error()
    assert L.wt_qk_rows_batch(*args(2, 64, 8)) == -1 and b"rows 8..18 of 16" in L.wt_last_error()
    assert L.wt_qk_rows_batch(*args(2, 32, 0)) == -3 and b"head_dim=32" in L.wt_last_error()
    # wt_logprob_digest_streams(logits, row_stride, n_rows, V, token, token_dtype, token_stride, ring_index, ring_rows, ring_row,
    #                           aux_tokens_host, n_aux, slice_begin, digest, slice, stream)
    aux = (C.c_int32 * 4)(1, 2, 3, 4)
    dg = lambda **kw: L.wt_logprob_digest_streams(*[kw.get(k, d) for k, d in (   # noqa: E731
        ("logits", p), ("row_stride", 100), ("n_rows", 4), ("V", 100), ("token", p), ("token_dtype", 1), ("token_stride", 1),
        ("ring_index", p), ("ring_rows", 8), ("ring_row", 0), ("aux", aux), ("n_aux", 2), ("slice_begin", 90), ("digest", p),
        ("slice", p), ("stream", 0))])
    assert dg(n_rows=0) == 0                                                           # no rows: nothing to do
    for bad in (dict(row_stride=10), dict(ring_row=8), dict(ring_rows=0), dict(n_aux=5), dict(slice_begin=100),
                dict(token_dtype=2), dict(ring_index=0), dict(digest=0), dict(token=0), dict(logits=0)):
        assert dg(**bad) == -1 and b"wt_logprob_digest_streams" in L.wt_last_error(), bad
    assert L.wt_shutdown() == 0


def test_transcribe_signature_is_the_reference_one():
    """Positional order and defaults of transcribe_timestamped (reference transcribe.py:79-120)."""
    import inspect
    import whisper_timestamped as wt
    expected = [("model", inspect._empty), ("audio", inspect._empty), ("language", None), ("task", "transcribe"),
                ("remove_punctuation_from_words", False), ("compute_word_confidence", True),
                ("include_punctuation_in_confidence", False), ("refine_whisper_precision", 0.5), ("min_word_duration", 0.02),
                ("plot_word_alignment", False), ("word_alignment_most_top_layers", None), ("remove_empty_words", False),
                ("use_backend_timestamps", False), ("seed", 1234), ("vad", False), ("detect_disfluencies", False),
                ("trust_whisper_timestamps", True), ("naive_approach", False), ("temperature", 0.0), ("best_of", None),
                ("beam_size", None), ("patience", None), ("length_penalty", None), ("compression_ratio_threshold", 2.4),
                ("logprob_threshold", -1.0), ("no_speech_threshold", 0.6), ("fp16", None),
                ("condition_on_previous_text", True), ("initial_prompt", None), ("suppress_tokens", "-1"),
                ("sample_len", None), ("verbose", False)]
    got = [(n, p.default) for n, p in inspect.signature(wt.transcribe_timestamped).parameters.items()]
    assert got == expected
    got = [(n, p.default) for n, p in inspect.signature(wt.load_model).parameters.items()]
    assert got == [("name", inspect._empty), ("device", None), ("backend", "openai-whisper"), ("download_root", None),
                   ("in_memory", False)]


def test_output_surface_matches_reference():
    """filtered_keys / flatten / remove_keys / write_csv against what the reference's own functions make of the
    reference's own results (tests/golden/output_surface.json, written by tests/golden/make_golden_output.py)."""
    import json
    from golden.make_golden_output import surface
    from whisper_timestamped import output
    here = os.path.dirname(os.path.abspath(__file__))
    golden = json.load(open(os.path.join(here, "golden", "output_surface.json"), encoding="utf-8"))
    cases = {c["name"]: c for c in json.load(open(os.path.join(here, "golden", "transcribe_cases.json"), encoding="utf-8"))}
    assert len(golden) == 5
    for name, exp in golden.items():
        got = json.loads(json.dumps(surface(output, cases[name]["expected"])))
        assert got == exp, name
    assert list(output.flatten([[1, 2], [3]])) == [1, 2, 3]
    assert list(output.flatten([{"a": [1]}, {}], "a")) == [1]


def test_small_unit_mirror_matches_the_header(tmp_path):
    """_lib.small_unit (Python, informational) against wt_small_unit (csrc/wt_small.h, what the library and its kernels
    decide with): the header is compiled into a host program that prints the decision for a grid of shapes."""
    import shutil
    import subprocess
    from whisper_timestamped import _lib
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    src = tmp_path / "probe.hip"
    src.write_text('#include <cstdio>\n#include "wt_small.h"\n'
                   'int main() { for (int T = 1; T <= 256; ++T) for (int F = 1; F <= 1792; F += (F < 40 ? 1 : 37))\n'
                   '  std::printf("%d %d %d %lld\\n", T, F, (int)wt::wt_small_unit(T, F), wt::wt_small_lds_bytes(T, F)); return 0; }\n')
    exe = tmp_path / "probe"
    csrc = os.path.join(ROOT, "whisper-timestamped_amd", "csrc")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-std=c++17", "-I" + csrc, "-I" + os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe)])      # (host program: nothing is launched, no GPU needed)
    n = small = 0
    for line in subprocess.check_output([str(exe)], text=True).splitlines():
        T, F, want, need = map(int, line.split())
        assert _lib.small_unit(T, F) == bool(want), (T, F, need)
        n += 1
        small += want
    assert n > 20000 and 0 < small < n
    assert _lib.small_unit(11, 144) and _lib.small_unit(64, 256) and not _lib.small_unit(224, 1500)


def test_rounding_helpers_equal_the_builtin_round_on_numpy_scalars():
    """words.round_timestamp / round_confidence: for numpy.float64 inputs the fast path must be bit-identical to
    round(numpy.float64, n) -- numpy's multiply / rint / divide -- including ties, negatives, the sign of zero, inf, nan;
    Python floats keep the builtin (correctly rounded) result, as in the reference (transcribe.py:1807-1811)."""
    from whisper_timestamped import words
    rng = np.random.RandomState(1)
    vals = np.concatenate([rng.rand(60000) * 3700, rng.rand(20000), np.round(rng.rand(40000) * 100, 3) + 0.005, -rng.rand(10000) * 10,
                           np.arange(0, 40, 0.005), np.array([0.0, -0.0, 1e-9, -1e-9, 1e15, 2.675, 1.005, 0.125, 0.375, np.inf, -np.inf, np.nan])])
    for x in vals:
        x = np.float64(x)
        for fn, nd in ((words.round_timestamp, 2), (words.round_confidence, 3)):
            a, b = round(x, nd), fn(x)
            assert type(b) is np.float64
            assert (a == b and np.signbit(a) == np.signbit(b)) or (np.isnan(a) and np.isnan(b)), (x, nd, a, b)
    for x in (2.675, 1.005, 0.125, 12.3456, -0.0049):            # Python floats: the builtin
        assert words.round_timestamp(x) == round(x, 2) and type(words.round_timestamp(x)) is float
        assert words.round_confidence(x) == round(x, 3)
