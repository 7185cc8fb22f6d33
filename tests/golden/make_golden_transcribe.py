#!/usr/bin/env python3
"""Generate tests/golden/transcribe_cases.json by running THE REFERENCE'S OWN
``transcribe_timestamped`` (/root/reference/whisper_timestamped/transcribe.py,
unmodified) in the build container, on the CPU, against

  * ``tests/whisper_double`` registered as the ``whisper`` package (the real
    openai-whisper is not installed; see that package's docstring), with
    random-initialised models and SCRIPTED sampling so that every branch of the
    hook state machine is reached deterministically;
  * a ``dtw`` stub backed by oracle/dtw_patterns.py, the generic step-pattern interpreter, cross-checked per call
    against oracle/dtw_ref.c (dtw-python is not installed).

The GPU tests (tests/test_gpu_transcribe.py) rebuild the same model / audio /
script from the case parameters and compare this repository's ``transcribe``
with the stored reference output.

Run:  python tests/golden/make_golden_transcribe.py      (needs /root/reference)
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

REF = "/root/reference/whisper_timestamped/transcribe.py"
OUT = os.path.join(HERE, "transcribe_cases.json")


# --------------------------------------------------------------------------- shared with the GPU tests
def text_ids(seed, n):
    """Text-token ids that no logit filter suppresses (the KAT pieces + synthetic ids)."""
    import whisper_double.tokenizer as T
    tk = T.get_tokenizer(True, language="en")
    banned = set(tk.non_speech_tokens) | {220}
    pool = [t for t in (6455, 2232, 286, 2041, 8660, 291, 808, 493, 365, 445, 718, 505, 458, 4666, 1022, 6992, 631, 7282,
                        1956, 871, 8208, 517, 5977, 7418) if t not in banned]
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        r = rng.rand()
        if r < 0.5:
            out.append(int(pool[rng.randint(len(pool))]))
        elif r < 0.9:
            t = int(rng.randint(300, 40000))
            out.append(t if t not in banned else 300)
        else:
            out.append(int([11, 13][rng.randint(2)]))      # "," "."
    return out


def window_script(ts0, eot, segments, ending="eot"):
    """segments: [(start_frame, [text ids], end_frame)].  Whisper's pattern: <|s|> text <|e|><|s'|> text <|e'|> ... ;
    ending: "eot" -> ... <|e|> eot ("single timestamp ending"); "pair" -> ... <|e|><|e|> eot (no speech after);
    "noend" -> text then eot (no closing timestamp); "limit" -> no eot at all (decoder hits sample_len)."""
    toks = []
    for k, (s, text, e) in enumerate(segments):
        toks.append(ts0 + s)
        toks.extend(text)
        last = k == len(segments) - 1
        if last and ending == "noend":
            break
        if last and ending == "limit":
            break
        toks.append(ts0 + e)
        if last and ending == "pair":
            toks.append(ts0 + e)
    if ending != "limit":
        toks.append(eot)
    return toks


def case_list():
    C = []
    ML, EN = 50364, 50363           # timestamp_begin
    EOT_ML, EOT_EN = 50257, 50256

    def seg(seed, s, n, e):
        """n text tokens between <|s|> and <|e|>: odd seeds script the ids (low probability under a random model),
        even seeds let the model pick its most likely text token (None) so that confidences are not all ~0."""
        return (s, text_ids(seed, n) if seed % 2 else [None] * n, e)

    C.append(dict(name="one_window_two_segments", model="tiny", audio_s=12.0, audio_seed=1,
                  opts=dict(language="en"),
                  script=[window_script(ML, EOT_ML, [seg(1, 10, 7, 180), seg(2, 200, 9, 520)], "eot")]))
    C.append(dict(name="two_windows_prompted", model="tiny", audio_s=47.0, audio_seed=2,
                  opts=dict(language="en"),
                  script=[window_script(ML, EOT_ML, [seg(3, 0, 8, 300), seg(4, 300, 6, 700), seg(5, 720, 10, 1300)], "pair"),
                          window_script(ML, EOT_ML, [seg(6, 25, 6, 400), seg(7, 410, 5, 800)], "eot")]))
    C.append(dict(name="eot_without_end_timestamp", model="tiny", audio_s=9.0, audio_seed=3,
                  opts=dict(language="en"),
                  script=[window_script(ML, EOT_ML, [seg(8, 5, 6, 150), seg(9, 160, 7, 0)], "noend")]))
    C.append(dict(name="decoding_limit", model="tiny", audio_s=20.0, audio_seed=4,
                  opts=dict(language="en", sample_len=40),
                  script=[window_script(ML, EOT_ML, [seg(10, 12, 9, 260), seg(11, 270, 40, 0)], "limit")[:40]]))
    C.append(dict(name="language_detection", model="tiny", audio_s=8.0, audio_seed=5,
                  opts=dict(language=None),
                  script=[window_script(ML, EOT_ML, [seg(12, 20, 8, 350)], "eot")]))
    C.append(dict(name="english_only_model", model="tiny.en", audio_s=10.0, audio_seed=6,
                  opts=dict(language="en"),
                  script=[window_script(EN, EOT_EN, [seg(13, 8, 6, 210), seg(14, 230, 8, 480)], "eot")]))
    C.append(dict(name="no_trust_whisper_timestamps", model="tiny", audio_s=36.0, audio_seed=7,
                  opts=dict(language="en", trust_whisper_timestamps=False),
                  script=[window_script(ML, EOT_ML, [seg(15, 6, 7, 280), seg(16, 300, 8, 690), seg(17, 700, 6, 1100)], "pair"),
                          window_script(ML, EOT_ML, [seg(18, 15, 9, 500)], "eot")]))
    C.append(dict(name="punctuation_options", model="tiny", audio_s=11.0, audio_seed=8,
                  opts=dict(language="en", include_punctuation_in_confidence=True, remove_punctuation_from_words=True),
                  script=[window_script(ML, EOT_ML, [(4, [6455, 11, 2232, 11, 286, 2041, 13], 240),
                                                     (250, [8660, 291, 808, 13, 13, 493], 500)], "eot")]))
    # (compute_word_confidence=False together with no_speech_threshold=None crashes the reference itself:
    #  chunk_logprobs stays empty, transcribe.py:526 -- so only the first is switched off here)
    C.append(dict(name="no_confidence", model="tiny", audio_s=7.0, audio_seed=9,
                  opts=dict(language="en", compute_word_confidence=False),
                  script=[window_script(ML, EOT_ML, [seg(19, 3, 9, 330)], "eot")]))
    C.append(dict(name="no_speech_skip", model="tiny", audio_s=40.0, audio_seed=10,
                  opts=dict(language="en", no_speech_threshold=1e-12, logprob_threshold=-0.05),
                  script=[window_script(ML, EOT_ML, [seg(20, 10, 6, 300), seg(21, 310, 7, 800)], "pair"),
                          window_script(ML, EOT_ML, [seg(22, 5, 5, 200)], "eot")]))
    C.append(dict(name="all_heads_top_layers", model="tiny", audio_s=9.0, audio_seed=11,
                  opts=dict(language="en", word_alignment_most_top_layers=2),
                  script=[window_script(ML, EOT_ML, [seg(23, 7, 8, 300)], "eot")]))
    C.append(dict(name="disfluencies_and_empty_words", model="tiny", audio_s=14.0, audio_seed=12,
                  opts=dict(language="en", detect_disfluencies=True, remove_empty_words=True, min_word_duration=0.04),
                  script=[window_script(ML, EOT_ML, [seg(24, 2, 12, 420), seg(25, 440, 10, 690)], "eot")]))
    C.append(dict(name="norefine_french", model="tiny", audio_s=10.0, audio_seed=13,
                  opts=dict(language="fr", refine_whisper_precision=0.0),
                  script=[window_script(ML, EOT_ML, [(5, [11771, 17134, 4666, 1022, 875, 2557, 68], 260),
                                                     (270, [6992, 631, 269, 6, 377, 409, 7282], 490)], "eot")]))
    C.append(dict(name="vad_explicit_islands", model="tiny", audio_s=30.0, audio_seed=18,
                  opts=dict(language="en", vad=[(2.0, 9.5), (14.0, 21.25)]),
                  script=[window_script(ML, EOT_ML, [seg(33, 5, 8, 330), seg(34, 340, 9, 700)], "eot")]))
    C.append(dict(name="initial_prompt_and_translate", model="tiny", audio_s=38.0, audio_seed=19,
                  opts=dict(language="fr", task="translate", initial_prompt="So, uh, I guess"),
                  script=[window_script(ML, EOT_ML, [seg(36, 4, 7, 320), seg(37, 330, 6, 900)], "pair"),
                          window_script(ML, EOT_ML, [seg(38, 10, 8, 450)], "eot")]))
    C.append(dict(name="english_only_no_trust_no_condition", model="tiny.en", audio_s=35.0, audio_seed=21,
                  opts=dict(language="en", trust_whisper_timestamps=False, condition_on_previous_text=False,
                            suppress_tokens="11,13"),
                  script=[window_script(EN, EOT_EN, [seg(44, 0, 6, 500), seg(46, 520, 7, 1200)], "pair"),
                          window_script(EN, EOT_EN, [seg(48, 20, 8, 400)], "eot")]))
    C.append(dict(name="language_detection_no_trust", model="tiny", audio_s=9.0, audio_seed=22,
                  opts=dict(language=None, trust_whisper_timestamps=False),
                  script=[window_script(ML, EOT_ML, [seg(50, 15, 7, 200), seg(52, 210, 6, 430)], "eot")]))
    C.append(dict(name="empty_first_window", model="tiny", audio_s=44.0, audio_seed=26,
                  opts=dict(language="en"),
                  script=[[ML + 10, EOT_ML],
                          window_script(ML, EOT_ML, [seg(67, 12, 7, 300), seg(68, 320, 8, 690)], "eot")]))
    C.append(dict(name="three_windows_single_segment_window", model="tiny", audio_s=75.0, audio_seed=27,
                  opts=dict(language="en", refine_whisper_precision=1.0, min_word_duration=0.1, remove_punctuation_from_words=True),
                  script=[window_script(ML, EOT_ML, [seg(69, 0, 9, 800)], "eot"),
                          window_script(ML, EOT_ML, [seg(70, 5, 6, 400), seg(71, 410, 7, 1000), seg(72, 1010, 5, 1450)], "pair"),
                          window_script(ML, EOT_ML, [seg(73, 30, 8, 600)], "noend")]))
    # large-v3's front end and vocabulary on a tiny-sized model: 128 mel bins, 100 languages, timestamps start at 50365
    C.append(dict(name="v3_like_128_mels", model="tiny-v3", audio_s=33.0, audio_seed=25,
                  opts=dict(language="en"),
                  script=[window_script(ML + 1, EOT_ML, [seg(64, 6, 8, 400), seg(65, 420, 7, 900)], "pair"),
                          window_script(ML + 1, EOT_ML, [seg(66, 10, 6, 300)], "eot")]))
    # ---- naive strategy (transcribe, then teacher-forced re-run) -------------------------------------
    C.append(dict(name="naive_greedy", model="tiny", audio_s=12.0, audio_seed=14,
                  opts=dict(language="en", naive_approach=True),
                  script=[window_script(ML, EOT_ML, [seg(26, 10, 7, 200), seg(27, 210, 8, 560)], "eot")]))
    C.append(dict(name="naive_beam", model="tiny", audio_s=41.0, audio_seed=15,
                  opts=dict(language="en", beam_size=2),
                  # (beam search is forced at the RESULT level: explicit ids only, i.e. odd seeds)
                  script=[window_script(ML, EOT_ML, [seg(41, 0, 6, 400), seg(29, 420, 7, 1100)], "pair"),
                          window_script(ML, EOT_ML, [seg(43, 12, 6, 380)], "eot")]))
    # BASELINE config 3: whisper-small multilingual, beam_size=5 (two-pass path).  The model carries no alignment_heads
    # attribute here, so the heads come from the parameter-count table (10 heads in layers 5..10).
    C.append(dict(name="small_beam5_table_heads", model="small", drop_alignment_heads=True, audio_s=14.0, audio_seed=24,
                  opts=dict(language="en", beam_size=5),
                  script=[window_script(ML, EOT_ML, [seg(61, 8, 7, 300), seg(63, 310, 6, 640)], "eot")]))
    C.append(dict(name="naive_no_trust", model="tiny", audio_s=13.0, audio_seed=16,
                  opts=dict(language="en", naive_approach=True, trust_whisper_timestamps=False,
                            include_punctuation_in_confidence=True),
                  script=[window_script(ML, EOT_ML, [seg(31, 5, 8, 250), (260, [6455, 11, 2232, 13], 600)], "eot")]))
    C.append(dict(name="naive_sampling_best_of", model="tiny", audio_s=11.0, audio_seed=23,
                  opts=dict(language="en", temperature=0.5, best_of=2),
                  script=[window_script(ML, EOT_ML, [seg(55, 6, 7, 260), seg(57, 270, 8, 520)], "eot")]))
    C.append(dict(name="naive_language_detection", model="tiny", audio_s=8.0, audio_seed=17,
                  opts=dict(language=None, temperature=(0.0, 0.4)),
                  script=[window_script(ML, EOT_ML, [seg(32, 6, 7, 330)], "eot")]))
    C.append(dict(name="naive_vad_islands", model="tiny", audio_s=28.0, audio_seed=28,
                  opts=dict(language="en", naive_approach=True, vad=[(1.0, 8.25), (12.5, 19.0)]),
                  script=[window_script(ML, EOT_ML, [seg(74, 4, 7, 300), seg(75, 320, 8, 660)], "eot")]))
    C.append(dict(name="naive_disfluencies_punctuation", model="tiny", audio_s=15.0, audio_seed=29,
                  opts=dict(language="en", naive_approach=True, detect_disfluencies=True, remove_punctuation_from_words=True,
                            min_word_duration=0.06),
                  script=[window_script(ML, EOT_ML, [(3, [6455, 11, 2232, 286, 13, 2041, 8660], 380),
                                                     seg(76, 400, 11, 730)], "eot")]))
    C.append(dict(name="naive_two_windows_prompted", model="tiny", audio_s=43.0, audio_seed=30,
                  opts=dict(language="en", naive_approach=True, initial_prompt="Well then"),
                  script=[window_script(ML, EOT_ML, [seg(77, 0, 7, 350), seg(78, 360, 6, 820), seg(79, 840, 8, 1380)], "pair"),
                          window_script(ML, EOT_ML, [seg(80, 12, 7, 420)], "eot")]))
    # ---- naive strategy, trust_whisper_timestamps=False over SEVERAL windows: the 30 s seek groups are independent
    #      (transcribe.py:1197-1202), which is what whisper_timestamped/batched.py runs as one batch -----------------
    C.append(dict(name="naive_no_trust_three_windows", model="tiny", audio_s=75.0, audio_seed=31,
                  opts=dict(language="en", naive_approach=True, trust_whisper_timestamps=False),
                  script=[window_script(ML, EOT_ML, [seg(81, 0, 8, 420), seg(82, 440, 7, 900), seg(83, 920, 6, 1400)], "pair"),
                          window_script(ML, EOT_ML, [seg(84, 10, 9, 600), seg(85, 620, 5, 1250)], "pair"),
                          window_script(ML, EOT_ML, [seg(86, 20, 7, 500)], "eot")]))
    C.append(dict(name="naive_no_trust_beam_two_windows", model="tiny", audio_s=52.0, audio_seed=32,
                  opts=dict(language="en", beam_size=2, trust_whisper_timestamps=False, include_punctuation_in_confidence=True),
                  script=[window_script(ML, EOT_ML, [seg(87, 0, 6, 500), (520, [6455, 11, 2232, 286, 13], 1100)], "pair"),
                          window_script(ML, EOT_ML, [seg(89, 15, 8, 700)], "eot")]))
    C.append(dict(name="naive_no_trust_disfluencies_padding", model="tiny", audio_s=37.0, audio_seed=33,
                  opts=dict(language="en", naive_approach=True, trust_whisper_timestamps=False, detect_disfluencies=True,
                            remove_punctuation_from_words=True, refine_whisper_precision=0.2),
                  script=[window_script(ML, EOT_ML, [(3, [6455, 11, 2232, 286, 13, 2041, 8660], 380),
                                                     seg(90, 400, 11, 930), seg(91, 950, 6, 1450)], "pair"),
                          window_script(ML, EOT_ML, [seg(92, 5, 7, 300)], "noend")]))
    C.append(dict(name="naive_no_trust_english_only_four_windows", model="tiny.en", audio_s=100.0, audio_seed=34,
                  opts=dict(language="en", naive_approach=True, trust_whisper_timestamps=False, compute_word_confidence=True,
                            condition_on_previous_text=False),
                  script=[window_script(EN, EOT_EN, [seg(93, 0, 7, 700), seg(94, 720, 8, 1490)], "pair"),
                          window_script(EN, EOT_EN, [seg(95, 0, 6, 800)], "pair"),
                          window_script(EN, EOT_EN, [seg(96, 30, 9, 600), seg(97, 610, 6, 1300)], "pair"),
                          window_script(EN, EOT_EN, [seg(98, 10, 5, 350)], "eot")]))
    # ---- the "peaked" double (whisper_double.model.sharpen_cross_attention): cross-attention with a monotone ridge on the
    #      alignment heads, as a trained model has, and scripts whose timestamps follow it (many_helper.peaked_segments) --
    #      the cases above run on plain random-init weights, whose flat attention leaves the DTW little to decide on
    import many_helper as H

    def pseg(seed, counts, first_pos=3):
        return [(s, text_ids(seed + k, n) if (seed + k) % 2 else [None] * n, e)
                for k, (s, n, e) in enumerate(H.peaked_segments(counts, first_pos))]

    C.append(dict(name="peaked_one_window", model="tiny", attention="peaked", audio_s=14.0, audio_seed=41,
                  opts=dict(language="en"),
                  script=[window_script(ML, EOT_ML, pseg(101, [9, 14, 7, 11, 16]), "eot")]))
    C.append(dict(name="peaked_two_windows_no_condition", model="tiny", attention="peaked", audio_s=49.0, audio_seed=42,
                  opts=dict(language="en", condition_on_previous_text=False),
                  script=[window_script(ML, EOT_ML, pseg(111, [12, 20, 9, 25, 14, 18, 30, 11]), "pair"),
                          window_script(ML, EOT_ML, pseg(121, [10, 8, 15]), "eot")]))
    C.append(dict(name="peaked_base_table_heads", model="base", attention="peaked", drop_alignment_heads=True, audio_s=22.0,
                  audio_seed=43, opts=dict(language="en"),
                  script=[window_script(ML, EOT_ML, pseg(131, [17, 22, 13, 28, 19, 24]), "eot")]))
    C.append(dict(name="peaked_full_window_disfluencies", model="tiny", attention="peaked", audio_s=30.0, audio_seed=44,
                  opts=dict(language="en", detect_disfluencies=True),
                  script=[window_script(ML, EOT_ML, pseg(141, [21, 35, 18, 40, 27, 33, 22]), "eot")]))
    C.append(dict(name="peaked_naive_no_trust", model="tiny", attention="peaked", audio_s=27.0, audio_seed=45,
                  opts=dict(language="en", naive_approach=True, trust_whisper_timestamps=False),
                  script=[window_script(ML, EOT_ML, pseg(151, [15, 26, 12, 31, 20]), "eot")]))
    C.append(dict(name="peaked_conditioned_second_window", model="tiny", attention="peaked", audio_s=52.0, audio_seed=46,
                  opts=dict(language="en"),
                  script=[window_script(ML, EOT_ML, pseg(161, [14, 19, 23, 12, 17]), "pair"),
                          window_script(ML, EOT_ML, pseg(171, [11, 16, 9]), "eot")]))
    return C


def build_case(c, device="cpu"):
    """-> (model, audio tensor on the CPU, Script)"""
    import whisper_double as W
    from whisper_double.decoding import Script
    model = W.build_model(c["model"], seed=c.get("model_seed", 0), device=device, attention=c.get("attention", "flat"))
    if c.get("drop_alignment_heads"):
        del model.alignment_heads
    g = torch.Generator().manual_seed(1000 + c["audio_seed"])
    n = int(round(c["audio_s"] * 16000))
    t = torch.arange(n) / 16000.0
    audio = 0.05 * torch.randn(n, generator=g) + 0.1 * torch.sin(2 * np.pi * 220.0 * t) * (torch.sin(2 * np.pi * 0.7 * t) > 0)
    return model, audio.float(), Script(c["script"])


def public_view(result):
    """What the parity test compares: the JSON surface (times, texts, confidences)."""
    out = dict(text=result["text"], language=result.get("language"), segments=[])
    if "speech_activity" in result:
        out["speech_activity"] = result["speech_activity"]
    if "language_probs" in result:
        lp = result["language_probs"]
        top = sorted(lp, key=lp.get, reverse=True)[:3]
        out["language_probs_top"] = {k: lp[k] for k in top}
    for s in result["segments"]:
        seg = {k: s[k] for k in ("id", "seek", "start", "end", "text", "tokens", "temperature", "avg_logprob",
                                 "compression_ratio", "no_speech_prob") if k in s}
        if "confidence" in s:
            seg["confidence"] = s["confidence"]
        seg["words"] = [dict(w) for w in s.get("words", [])]
        out["segments"].append(seg)
    return out


def raw_confidences(view):
    """Flat list: per segment its confidence (if any), then its words' confidences."""
    out = []
    for s in view["segments"]:
        if "confidence" in s:
            out.append(s["confidence"])
        out.extend(w["confidence"] for w in s["words"] if "confidence" in w)
    return out


# --------------------------------------------------------------------------- reference loading
def load_reference():
    import whisper_double as W
    from oracle import align_ref as O
    W.install()
    from oracle import dtw_patterns as P

    def c_restatement(x, pattern):
        r = O.dtw_ref(x, step_pattern=0 if pattern.n_patterns == 3 else 1)
        return r.index1s, r.index2s
    sys.modules.update(P.stub_modules(cross_check=c_restatement))
    spec = importlib.util.spec_from_file_location("ref_transcribe", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    from whisper_double.decoding import set_script
    ref = load_reference()
    only = set(sys.argv[1:])
    cases = []
    if only and os.path.exists(OUT):
        cases = json.load(open(OUT))
    done = {c["name"]: c for c in cases}
    for c in case_list():
        if only and c["name"] not in only:
            continue
        model, audio, script = build_case(c)
        set_script(script)
        try:
            result = ref.transcribe_timestamped(model, audio, fp16=False, **c["opts"])
        finally:
            set_script(None)
        rec = dict(c)
        rec["expected"] = json.loads(json.dumps(public_view(result), default=float))
        rec["recorded"] = script.record
        # the same run with the reference's round_confidence (transcribe.py:1807) switched off: confidences before
        # round(, 3), for the 1e-4 bar on the raw value
        from whisper_double.decoding import Script
        model, audio, _ = build_case(c)
        set_script(Script(script.record))
        keep = ref.round_confidence
        ref.round_confidence = lambda x: x
        try:
            raw = ref.transcribe_timestamped(model, audio, fp16=False, **c["opts"])
        finally:
            ref.round_confidence = keep
            set_script(None)
        rec["expected_raw_confidence"] = raw_confidences(json.loads(json.dumps(public_view(raw), default=float)))
        done[c["name"]] = rec
        nw = sum(len(s["words"]) for s in rec["expected"]["segments"])
        print(f"{c['name']:32s} segments={len(rec['expected']['segments'])} words={nw} windows={len(script.record)}")
    order = [c["name"] for c in case_list()]
    with open(OUT, "w", encoding="utf-8") as f:
        json.dump([done[n] for n in order if n in done], f, ensure_ascii=False, indent=0)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
