"""plot_word_alignment: the debug figures of whisper_timestamped/plotting.py (reference transcribe.py:1586-1646,
1685-1700, 1756-1781, 2139-2150).  A figure is not a result: what is pinned is that one is produced per aligned segment,
under the reference's file names, and that asking for figures changes nothing in what transcribe() returns."""
import glob
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import matplotlib
matplotlib.use("Agg")


def _is_jpeg(path):
    with open(path, "rb") as f:
        return f.read(3) == b"\xff\xd8\xff"


def _synthetic_unit(T=7, F=120, disfluencies=True, mel=True):
    rng = np.random.RandomState(0)
    stairs = np.sort(rng.randint(0, F, size=T - 1))
    jumps = np.concatenate([[0], stairs, [F - 1]]).astype(np.int64)
    attention = rng.rand(T, F) * 0.05
    for i in range(T):
        attention[i, jumps[i]:max(jumps[i + 1], jumps[i] + 1)] += 0.5 + 0.3 * np.sin(np.arange(max(jumps[i + 1] - jumps[i], 1)) / 2.0)
    path_tokens = np.repeat(np.arange(T), np.maximum(np.diff(jumps), 1))
    path_frames = np.arange(len(path_tokens))
    unit = SimpleNamespace(word_pieces=[["<|0.00|>"], [" he", "llo"], [" wor", "ld", "!"], ["<|2.40|>"]], start_token=10,
                           end_token=10 + F, detect_disfluencies=disfluencies,
                           mel=torch.randn(1, 80, 3000) if mel else None)
    t0 = 10 * 0.02
    words = [dict(text="hello", start=t0 + 0.02 * jumps[1], end=t0 + 0.02 * jumps[3], tokens=[" he", "llo"], tokens_indices=[1, 2]),
             dict(text="[*]", start=t0 + 0.02 * jumps[3], end=t0 + 0.02 * (jumps[3] + 4), tokens=[], tokens_indices=[]),
             dict(text="world!", start=t0 + 0.02 * (jumps[3] + 4), end=t0 + 0.02 * jumps[6], tokens=[" wor", "ld", "!"],
                  tokens_indices=[3, 4, 5])]
    return unit, -attention, path_tokens, path_frames, jumps, words


@pytest.mark.parametrize("disfluencies,mel", [(True, True), (False, True), (True, False), (False, False)])
def test_alignment_figures_are_numbered_per_call_and_saved_as_jpeg(tmp_path, disfluencies, mel):
    from whisper_timestamped import plotting
    unit, cost, pt, pf, jumps, words = _synthetic_unit(disfluencies=disfluencies, mel=mel)
    prefix = str(tmp_path / "clip.wav")
    plotting.reset()
    plotting.alignment_figure(unit, cost, pt, pf, jumps, words, prefix)
    plotting.alignment_figure(unit, cost, pt, pf, jumps, words, prefix)
    assert sorted(os.path.basename(p) for p in glob.glob(prefix + ".*")) == ["clip.wav.alignment001.jpg", "clip.wav.alignment002.jpg"]
    assert all(_is_jpeg(p) for p in glob.glob(prefix + ".*"))
    plotting.reset()                                     # the next transcribe() call starts at 001 again (T.py:300-301)
    plotting.alignment_figure(unit, cost, pt, pf, jumps, words, str(tmp_path / "other"))
    assert os.path.exists(str(tmp_path / "other.alignment001.jpg"))
    import matplotlib.pyplot as plt
    assert plt.get_fignums() == []                       # saved figures are closed


def test_vad_figure(tmp_path):
    from whisper_timestamped import plotting
    audio = torch.randn(16000 * 7)
    plotting.vad_figure(audio, [(1000, 30000), (50000, 100000)], 16000, str(tmp_path / "rec"))
    assert _is_jpeg(str(tmp_path / "rec.VAD.jpg"))


def test_remove_non_speech_draws_the_islands_when_asked(tmp_path):
    from whisper_timestamped import vad
    audio = torch.randn(16000 * 12)
    a, segs, convert = vad.remove_non_speech(audio, method=[(1.0, 4.0), (6.0, 9.5)], plot=str(tmp_path / "rec"))
    b, segs_b, _ = vad.remove_non_speech(audio, method=[(1.0, 4.0), (6.0, 9.5)])
    assert _is_jpeg(str(tmp_path / "rec.VAD.jpg")) and torch.equal(a, b) and segs == segs_b


def _double(device):
    import whisper_double as W
    W.install()
    return W, W.build_model("tiny", seed=0, device=device)


@pytest.mark.gpu
def test_perform_word_alignment_with_a_figure_returns_the_same_words(tmp_path):
    """The seam function on the GPU, on the reference-generated alignment fixtures: plot="<prefix>" writes
    <prefix>.alignment<NNN>.jpg from the unit's cost matrix and warping path (read back from the device) and returns what
    plot=False returns."""
    import json
    from golden.make_golden import build_case_inputs
    import whisper_timestamped as wt
    from whisper_timestamped import plotting
    cases = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "align_cases.json"), encoding="utf-8"))
    plotting.reset()
    drawn_figures = 0
    for case in cases[:12]:
        tokens, att, heads, mfcc, tok = build_case_inputs(case)
        kw = dict(use_space=case.get("use_space", True), mfcc=None if mfcc is None else mfcc.cuda(),
                  refine_whisper_precision_nframes=case["refine"], remove_punctuation_from_words=case.get("remove_punct", False),
                  alignment_heads=None if heads is None else np.array(heads), detect_disfluencies=case.get("disfl", False),
                  subwords_can_be_empty=not case.get("noempty", False))
        plain = wt.perform_word_alignment(tokens, [a.cuda() for a in att], tok, **kw)
        drawn = wt.perform_word_alignment(tokens, [a.cuda() for a in att], tok, plot=str(tmp_path / "seg"), **kw)
        assert drawn == plain, case.get("name")
        if plain or os.path.exists(str(tmp_path / f"seg.alignment{drawn_figures + 1:03d}.jpg")):
            drawn_figures += 1
            assert _is_jpeg(str(tmp_path / f"seg.alignment{drawn_figures:03d}.jpg")), case.get("name")
    assert drawn_figures >= 6


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [{}, {"naive_approach": True}, {"trust_whisper_timestamps": False},
                                   {"detect_disfluencies": True, "vad": [(0.5, 9.0), (11.0, 19.0)]}],
                         ids=["efficient", "naive", "no_trust", "disfluencies_vad"])
def test_transcribe_with_figures_returns_what_it_returns_without(tmp_path, extra):
    """transcribe(..., plot_word_alignment="<prefix>") on the GPU: one figure per aligned segment, numbered from 001, and
    the dictionary transcribe() returns is the one it returns without figures."""
    import golden.make_golden_transcribe as G
    from whisper_double.decoding import Script, set_script
    import whisper_timestamped as wt
    W, model = _double("cuda")
    ML, EOT = 50364, 50257
    windows = [G.window_script(ML, EOT, [(0, G.text_ids(3, 6), 250), (270, G.text_ids(4, 5), 520), (540, G.text_ids(5, 7), 780)], "eot")
               for _ in range(3)]
    g = torch.Generator().manual_seed(9)
    audio = (0.05 * torch.randn(20 * 16000, generator=g)).float()
    opts = dict(language="en", fp16=False, **extra)

    def run(**kw):
        set_script(Script(windows))
        try:
            return wt.transcribe(model, audio, **opts, **kw)
        finally:
            set_script(None)
    def same(a, b, key=""):
        # texts, tokens and TIMES identical; the backend's own floats (avg_logprob, no_speech_prob, confidences) move in
        # their sixth digit from run to run of the SAME call (its GEMMs): compared at 1e-4
        if isinstance(a, dict):
            return a.keys() == b.keys() and all(same(a[k], b[k], k) for k in a)
        if isinstance(a, (list, tuple)):
            return len(a) == len(b) and all(same(x, y, key) for x, y in zip(a, b))
        if isinstance(a, (float, np.floating)):
            return abs(float(a) - float(b)) <= (0.0 if key in ("start", "end") else 1e-4)
        return a == b
    plain = run()
    drawn = run(plot_word_alignment=str(tmp_path / "rec"))
    assert same(drawn, plain)
    figs = sorted(os.path.basename(p) for p in glob.glob(str(tmp_path / "rec.alignment*.jpg")))
    n_aligned = sum(1 for s in plain["segments"] if s.get("words"))
    # one figure per call of the alignment seam: per segment, or per 30 s window when whisper's segment times are not
    # trusted (the window's segments are then aligned as ONE unit, transcribe.py:1197-1202)
    n_figures = 1 if extra.get("trust_whisper_timestamps") is False else n_aligned
    assert n_aligned == 3 and figs == [f"rec.alignment{k + 1:03d}.jpg" for k in range(n_figures)]
    assert all(_is_jpeg(str(tmp_path / f)) for f in figs)
    assert os.path.exists(str(tmp_path / "rec.VAD.jpg")) == ("vad" in extra)
