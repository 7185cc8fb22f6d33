"""whisper_timestamped.transcribe_batch (streams.py): B independent recordings stepping through the decoder together,
against the REFERENCE'S OWN OUTPUT for each of them (tests/golden/transcribe_cases.json: what /root/reference's
transcribe_timestamped produced, one recording at a time).

No GPU here: kernels = the CPU oracle (tests/cpu_kernel_standin.py), so what is checked is the host side of the B-stream
form -- the lock-step window driver (openai-whisper's transcribe() loop restated per stream), the recorded-calls replay
into every stream's EfficientSession, the shared rings and the one-launch-set-per-window-set sink -- on every branch the
goldens reach.  The same tests run with the real kernels in tests/test_gpu_streams_batch.py.
"""
import copy
import json

import pytest
import torch

import cpu_kernel_standin
from golden import make_golden_transcribe as G
from test_transcribe_host import CASES, compare, raw_confidence_gap, raw_logprob_gap, rounded

BATCHABLE = [c for c in CASES if not (c["opts"].get("naive_approach") or c["opts"].get("beam_size") or c["opts"].get("best_of")
                                      or isinstance(c["opts"].get("temperature"), (list, tuple)) or c["opts"].get("vad"))]


def install_streams_standin(monkeypatch):
    """The oracle-backed stand-in for the one kernel entry streams.py adds to what cpu_kernel_standin covers."""
    from whisper_timestamped import streams

    def write_qk(self, q_layers, k_layers, ring_index, row):
        sl, sh, ss = (t.tolist() for t in self.sel)
        scale = 64 ** -0.25
        for i, h, s in zip(sl, sh, ss):
            l = self.used[i]
            q = q_layers[l][:, -1:, h * 64:(h + 1) * 64] * scale
            k = k_layers[l][:, :, h * 64:(h + 1) * 64] * scale
            self.qk[ring_index.long(), s, row] = (q @ k.transpose(1, 2))[:, 0].float().to(self.qk.dtype)
    monkeypatch.setattr(streams.StreamRings, "write_qk", write_qk)

    def write_digest(self, rows, tokens, ring_index, step):
        """wt_logprob_digest_streams restated with torch CPU ops (include/wtalign.h): [0] log_softmax(row)[token],
        [1] max, [2] log(sum(exp(row - max))), [3] argmax bits, [4..] raw logits of the aux tokens; the timestamp slice."""
        import numpy as np
        x = rows.float()
        idx = ring_index.long()
        m = x.max(dim=1).values
        rec = torch.full((x.shape[0], 8), float("-inf"))
        rec[:, 0] = torch.log_softmax(x, dim=-1)[torch.arange(x.shape[0]), tokens.long()]
        rec[:, 1] = m
        rec[:, 2] = torch.logsumexp(x, dim=1) - m
        rec[:, 3] = torch.from_numpy(x.argmax(dim=1).to(torch.int32).numpy().view(np.float32).copy())
        for k, t in enumerate(self.aux_tokens):
            rec[:, 4 + k] = x[:, t]
        self.digest[idx, step] = rec
        self.slice[idx, step] = x[:, self.slice_begin:]
    monkeypatch.setattr(streams.StreamRings, "write_digest", write_digest)


def run_batch(cases, device="cpu", raw_confidence=True, max_streams=32, **extra):
    """transcribe_batch on the recordings of `cases` (same model, same options), every stream's sampler steered by its
    own recorded script.  -> list of public views."""
    import whisper_double as W
    from whisper_double.decoding import Script, set_row_scripts, set_script
    W.install()
    import whisper_timestamped as wt
    from whisper_timestamped import streams, words
    model = audios = None
    audios = []
    for c in cases:
        m, audio, _ = G.build_case(c, device=device)
        model = model or m
        audios.append(audio)
    scripts = [Script(c["recorded"]) for c in cases]

    def on_group(idx):                                   # idx: the recordings of this decoder loop, in row order
        for i in idx:
            scripts[i].begin_window()
        set_row_scripts([scripts[i] for i in idx])
    set_script(None)
    streams.ON_GROUP_DECODE = on_group
    words.RAW_CONFIDENCE = bool(raw_confidence)
    try:
        results = wt.transcribe_batch(model, audios, fp16=False, max_streams=max_streams, **cases[0]["opts"], **extra)
    finally:
        words.RAW_CONFIDENCE = False
        streams.ON_GROUP_DECODE = None
        set_row_scripts(None)
    for c, sc in zip(cases, scripts):
        assert sc.record == c["recorded"], c["name"]
    return [json.loads(json.dumps(G.public_view(r), default=float)) for r in results]


@pytest.mark.parametrize("case", BATCHABLE, ids=[c["name"] for c in BATCHABLE])
def test_a_batch_of_one_stream_equals_the_reference(case, monkeypatch):
    cpu_kernel_standin.install(monkeypatch)
    install_streams_standin(monkeypatch)
    raw = run_batch([copy.deepcopy(case)])[0]
    compare(rounded(raw), case["expected"], time_tol=0.0, conf_tol=0.0, logprob_tol=1e-6)
    assert raw_confidence_gap(raw, case) <= 1e-5 and raw_logprob_gap(raw, case) <= 2e-5


def same_model_cases():
    names = ("one_window_two_segments", "two_windows_prompted", "eot_without_end_timestamp", "empty_first_window")
    return [copy.deepcopy(next(c for c in CASES if c["name"] == n)) for n in names]


def test_streams_of_different_lengths_each_equal_the_reference(monkeypatch):
    """Four recordings of 9 to 47 s (one or two windows, different prompts in the second round) + the same four again:
    8 streams, ragged rounds, groups by prompt length -- every stream must reproduce the reference's output for it."""
    from whisper_timestamped import streams
    cpu_kernel_standin.install(monkeypatch)
    install_streams_standin(monkeypatch)
    cases = same_model_cases() + same_model_cases()[::-1]
    views = run_batch(cases)
    for raw, case in zip(views, cases):
        compare(rounded(raw), case["expected"], time_tol=0.0, conf_tol=0.0, logprob_tol=1e-5)
        assert raw_confidence_gap(raw, case) <= 1e-5 and raw_logprob_gap(raw, case) <= 2e-5
    assert streams.LAST_RUN["streams"] == 8 and streams.LAST_RUN["rounds"] >= 2
    # ONE alignment launch set per round (+ the final one), not one per stream and window
    assert streams.LAST_RUN["alignment_launch_sets"] <= streams.LAST_RUN["rounds"] + 1


def test_language_detection_for_every_stream_in_one_call(monkeypatch):
    """language=None: the backend's detect_language runs ONCE for all streams (one (B, 1) decoder call), every stream's
    session sees it as its own hooks would have (language probabilities, the no-speech probability the reference
    takes from that call) -- two streams, each equal to the reference's output for the recording."""
    cpu_kernel_standin.install(monkeypatch)
    install_streams_standin(monkeypatch)
    case = next(c for c in CASES if c["name"] == "language_detection")
    cases = [copy.deepcopy(case), copy.deepcopy(case)]
    for raw in run_batch(cases):
        compare(rounded(raw), case["expected"], time_tol=0.0, conf_tol=0.0, logprob_tol=1e-5)
        assert "language_probs_top" in raw and raw["language"] == case["expected"]["language"]


def test_more_recordings_than_streams(monkeypatch):
    """Continuous admission: fewer ring blocks than recordings -- a recording that is finished hands its block to the next
    one while the others go on (their tails are flushed before the block is overwritten).  Eight ragged recordings through
    two and through three blocks: each equals the reference's output for it."""
    from whisper_timestamped import streams
    cpu_kernel_standin.install(monkeypatch)
    install_streams_standin(monkeypatch)
    for n_blocks in (2, 3):
        cases = same_model_cases() + same_model_cases()[::-1]
        views = run_batch(cases, max_streams=n_blocks)
        for raw, case in zip(views, cases):
            compare(rounded(raw), case["expected"], time_tol=0.0, conf_tol=0.0, logprob_tol=1e-5)
        assert streams.LAST_RUN["ring_blocks"] == n_blocks and streams.LAST_RUN["streams"] == 8
        assert streams.LAST_RUN["admissions"] >= 3                    # blocks were handed on, more than once


def test_calls_the_batched_path_cannot_take_run_one_by_one(monkeypatch):
    """beam search / temperature fallback / vad belong to the naive strategy or pre-processing: transcribe_batch hands
    them to transcribe_timestamped, recording by recording (same results, no batching)."""
    import whisper_double as W
    W.install()
    import whisper_timestamped as wt
    from whisper_timestamped import streams
    assert not streams.supports(dict(temperature=(0.0, 0.2)), None, False)
    assert not streams.supports(dict(temperature=0.0, beam_size=5), None, True)
    assert not streams.supports(dict(temperature=0.0), [(0.0, 1.0)], False)
    assert streams.supports(dict(temperature=0.0, best_of=None, beam_size=None), None, False)
    with pytest.raises(AssertionError, match="unknown options"):
        wt.transcribe_batch(None, [torch.zeros(16000)], not_an_option=1)
    assert wt.transcribe_batch(None, []) == []
    assert streams.backend_missing() == []                       # the double has openai-whisper's DecodingTask surface
    real = W.decoding.DecodingTask._main_loop
    monkeypatch.delattr(W.decoding.DecodingTask, "_main_loop")
    assert streams.backend_missing() == ["DecodingTask._main_loop"]
    monkeypatch.setattr(W.decoding.DecodingTask, "_main_loop", real, raising=False)


def test_batched_timestamp_rules_equal_the_backends_row_by_row(monkeypatch):
    """streams.BatchedTimestampRules against the backend's ApplyTimestampRules on random token histories -- no
    timestamps yet, open pairs, closed pairs, text after a pair, the first sampled position, rows of one batch in
    different states: identical logits (values and -inf pattern)."""
    import numpy as np
    import whisper_double as W
    from whisper_timestamped import streams
    W.install()
    tk = W.tokenizer.get_tokenizer(True, language="en", task="transcribe")
    ts0, V = tk.timestamp_begin, 51865
    rng = np.random.RandomState(5)
    g = torch.Generator().manual_seed(5)
    for sample_begin, max_initial in ((4, 50), (4, None), (9, 50)):
        theirs = W.decoding.ApplyTimestampRules(tk, sample_begin, max_initial)
        mine = streams.BatchedTimestampRules.like(theirs)
        for n in (0, 1, 2, 3, 7, 30):
            rows = []
            for b in range(9):
                seq, t = [], int(rng.randint(0, 100))
                while len(seq) < n:
                    kind = rng.randint(4) if seq else 0
                    if kind == 0:                       # an opening timestamp
                        t += int(rng.randint(0, 40))
                        seq.append(ts0 + t)
                    elif kind == 1 and seq[-1] >= ts0:  # its twin
                        seq.append(seq[-1] if rng.rand() < 0.5 else ts0 + t + int(rng.randint(0, 9)))
                    else:
                        seq.append(int(rng.randint(300, 40000)))
                rows.append([50258, 50259, 50359, ts0 - 1][:min(4, sample_begin)] + [11] * (sample_begin - 4) + seq[:n])
            tokens = torch.tensor(rows)
            logits = torch.randn((len(rows), V), generator=g) * 3
            logits[:, ts0:] += float(rng.choice([-4.0, 0.0, 4.0]))        # both outcomes of the timestamp-mass rule
            a, b = logits.clone(), logits.clone()
            theirs.apply(a, tokens)
            mine.apply(b, tokens)
            assert torch.equal(torch.isinf(a), torch.isinf(b)), (sample_begin, n)
            assert torch.equal(a, b), (sample_begin, n)


def test_batched_suppress_tokens_equal_the_backends():
    import whisper_double as W
    from whisper_timestamped import streams
    W.install()
    theirs = W.decoding.SuppressTokens([3, 17, 50257, 220, 11])
    mine = streams.BatchedSuppressTokens(theirs.suppress_tokens)
    g = torch.Generator().manual_seed(1)
    a = torch.randn((5, 51865), generator=g)
    b = a.clone()
    theirs.apply(a, None)
    mine.apply(b, None)
    assert torch.equal(a, b) and int(torch.isinf(b).sum()) == 5 * 5


def test_window_bookkeeping_equals_the_backends_loop_on_random_token_streams():
    """streams._Stream.take_result restates what the backend's transcribe() loop does with the DecodingResult of a window
    (no-speech skip, segments at consecutive timestamps, seek, prompt bookkeeping).  The one-stream path runs the
    backend's own loop, the B-stream path this restatement: held against each other on random results -- no timestamps at
    all, single timestamps, pairs, every ending, empty results, no-speech windows -- through the backend's loop with a
    model whose decode() returns the scripted results."""
    import types

    import numpy as np
    import whisper_double as W
    from whisper_timestamped import streams
    W.install()
    tk = W.tokenizer.get_tokenizer(True, language="en", task="transcribe")
    ts0, eot = tk.timestamp_begin, tk.eot
    rng = np.random.RandomState(17)

    def random_result():
        kind = rng.randint(7)
        toks, t = [], int(rng.randint(0, 200))
        text = lambda n: [int(x) for x in rng.randint(300, 20000, size=n)]      # noqa: E731
        if kind == 0:
            toks = text(rng.randint(0, 12))                                     # no timestamps (or nothing at all)
        elif kind == 1:
            toks = [ts0 + t] + text(rng.randint(1, 9))                          # a start, no end
        else:
            for _ in range(rng.randint(1, 5)):
                e = t + int(rng.randint(1, 300))
                toks += [ts0 + t] + text(rng.randint(0, 8)) + [ts0 + min(e, 1500)]
                t = min(e + int(rng.randint(0, 30)), 1500)
            if kind == 2:
                toks = toks[:-1]                                                # the last segment has no closing timestamp
            elif kind == 3:
                toks.append(toks[-1])                                           # ... <|e|><|e|>: "no speech after"
            elif kind == 4:
                toks += [ts0 + t] + text(3)                                     # an open segment after the pairs
        return W.DecodingResult(audio_features=None, language="en", tokens=toks, text=tk.decode(toks).strip(),
                                avg_logprob=float(-rng.rand() * 1.5), no_speech_prob=float(rng.rand()),
                                temperature=0.0, compression_ratio=1.0)

    for trial in range(60):
        seconds = float(rng.choice([7.0, 30.0, 47.3, 95.0]))
        audio = torch.zeros(int(seconds * 16000))
        results = [random_result() for _ in range(40)]
        cond = bool(rng.randint(2))
        thresholds = dict(no_speech_threshold=[None, 0.6][rng.randint(2)], logprob_threshold=[None, -1.0][rng.randint(2)])
        used = []

        def decode(segment, options, _r=results, _u=used):
            _u.append(list(options.prompt or []))
            return _r[len(_u) - 1]
        model = types.SimpleNamespace(dims=types.SimpleNamespace(n_mels=80, n_audio_ctx=1500), is_multilingual=True, num_languages=99,
                                      device=torch.device("cpu"), decode=decode)
        want = W.transcribe(model, audio, temperature=0.0, condition_on_previous_text=cond, language="en", fp16=False,
                            compression_ratio_threshold=None, **thresholds)
        # the restatement, fed the same results in the same order
        mel = W.log_mel_spectrogram(audio, 80, padding=480000)
        st = streams._Stream(0, mel, mel.shape[-1] - 3000, None, tk, "en", [])
        opts = dict(condition_on_previous_text=cond, **thresholds)
        prompts, k = [], 0
        while st.active():
            st.window_mel()
            prompts.append(st.all_tokens[st.prompt_reset_since:])
            r = results[k]
            k += 1
            st.take_result(list(r.tokens), r.avg_logprob, r.no_speech_prob, 0.0, opts, W)
        got = st.transcription()
        assert k == len(used) and prompts == used, trial
        assert got["text"] == want["text"] and len(got["segments"]) == len(want["segments"]), trial
        for a, b in zip(got["segments"], want["segments"]):
            a = dict(a, compression_ratio=b["compression_ratio"])              # (computed from the text on both sides)
            assert a == b, (trial, a, b)


@pytest.mark.parametrize("seed,extra_opts", [(2024, None), (7, {"trust_whisper_timestamps": False}), (11, {"detect_disfluencies": True})],
                         ids=["defaults", "no_trust", "disfluencies"])
def test_random_scripts_streams_equal_one_stream_at_a_time(monkeypatch, seed, extra_opts, device="cpu", time_tol=0.0, n_wins=None):
    """Beyond the goldens: recordings with RANDOM scripted transcripts -- one to three windows, segments of random sizes,
    every ending the decoder can produce (closing timestamp, timestamp pair, no closing timestamp, token budget hit) --
    through transcribe_batch (three ring blocks for eight recordings: continuous admission, ragged rounds, groups by
    prompt length) and through transcribe() one at a time: identical dictionaries (oracle kernels on both sides)."""
    import numpy as np
    import whisper_double as W
    from whisper_double.decoding import Script, set_row_scripts, set_script
    W.install()
    import whisper_timestamped as wt
    from whisper_timestamped import streams, words
    if device == "cpu":
        cpu_kernel_standin.install(monkeypatch)
        install_streams_standin(monkeypatch)
    monkeypatch.setattr(words, "RAW_CONFIDENCE", True)     # (before the reference's round(, 3): a rounding flip is not a difference)
    rng = np.random.RandomState(seed)
    ML, EOT = 50364, 50257
    model = W.build_model("tiny", seed=0, device=device)

    def random_window(last):
        segs, t = [], int(rng.randint(0, 50))              # (the first timestamp of a window is at most 1.0 s: the filters)
        for _ in range(rng.randint(1, 5)):
            e = min(t + int(rng.randint(40, 420)), 1480)
            n = int(rng.randint(1, 9))
            segs.append((t, G.text_ids(int(rng.randint(1000)), n) if rng.rand() < 0.5 else [None] * n, e))
            t = min(e + int(rng.randint(0, 25)), 1490)
            if t >= 1480:
                break
        ending = ["eot", "pair", "noend", "eot"][rng.randint(4)] if last else ["eot", "pair"][rng.randint(2)]
        return G.window_script(ML, EOT, segs, ending)

    recs = []
    for k in range(8 if n_wins is None else len(n_wins)):
        n_win = int(rng.randint(1, 4)) if n_wins is None else n_wins[k]
        seconds = 30.0 * (n_win - 1) + float(rng.uniform(6.0, 29.0))
        g = torch.Generator().manual_seed(500 + k)
        audio = (0.05 * torch.randn(int(seconds * 16000), generator=g)).float()
        recs.append((audio, [random_window(j == n_win - 1) for j in range(n_win + 2)]))     # (+ spare windows, should the seek not reach the end)
    opts = dict(language="en", fp16=False, **(extra_opts or {}))
    singles, recorded = [], []
    for audio, windows in recs:
        sc = set_script(Script(windows))
        try:
            singles.append(wt.transcribe(model, audio, **opts))
        finally:
            set_script(None)
        recorded.append(sc.record)
    scripts = [Script(r) for r in recorded]                 # replay exactly what each one-stream run sampled

    def on_group(idx):
        for i in idx:
            scripts[i].begin_window()
        set_row_scripts([scripts[i] for i in idx])
    streams.ON_GROUP_DECODE = on_group
    try:
        batch = wt.transcribe_batch(model, [a for a, _ in recs], max_streams=3, **opts)
    finally:
        streams.ON_GROUP_DECODE = None
        set_row_scripts(None)
    assert streams.LAST_RUN["ring_blocks"] == 3 and (streams.LAST_RUN["admissions"] >= 3 or n_wins is not None)
    assert sum(streams.LAST_RUN["streams_per_loop"]) >= len(recs) and len(streams.LAST_RUN["streams_per_loop"]) == streams.LAST_RUN["decoder_loops"]
    n_words = 0
    for b, s_, sc, rec in zip(batch, singles, scripts, recorded):
        assert sc.record == rec
        vb, vs = (json.loads(json.dumps(G.public_view(x), default=float)) for x in (b, s_))
        compare(vb, vs, time_tol=time_tol, conf_tol=2e-5, logprob_tol=1e-4 if device != "cpu" else 1e-5)     # (GEMM batch-size rounding)
        n_words += sum(len(x["words"]) for x in vb["segments"])
    assert n_words > (40 if n_wins is None else 10)


def test_stuck_decoder_windows_with_a_prompt_derived_fallback_token(monkeypatch):
    """window A hits the decoding limit ending on TEXT, window B hits it too and is skipped as no-speech, so the
    prompt of window C still ends with A's last token -- the reference then reads the log-probability of an ARBITRARY text
    token from a window's last row (T.py:498-503).  The B-stream driver keeps that one row whole (StreamRings.last_row):
    transcribe_batch must give what transcribe() gives, not abort the batch."""
    import whisper_double as W
    from whisper_double.decoding import Script, set_row_scripts, set_script
    W.install()
    import whisper_timestamped as wt
    from whisper_timestamped import streams, words
    cpu_kernel_standin.install(monkeypatch)
    install_streams_standin(monkeypatch)
    monkeypatch.setattr(words, "RAW_CONFIDENCE", True)
    ML, EOT = 50364, 50257
    model = W.build_model("tiny", seed=0, device="cpu")
    limit = 24
    # one open segment each, no closing timestamp, no <|endoftext|>: the token budget ends them ON TEXT (a window with a
    # timestamp pair would close at the pair and drop its tail instead)
    a = [ML + 5] + [None] * (limit - 1)                                # the model's most likely tokens: kept
    b = [ML + 8] + G.text_ids(301, limit - 1)                          # unlikely ids: low avg log-prob -> skipped as no-speech
    c = G.window_script(ML, EOT, [(10, [None] * 7, 300), (320, [None] * 6, 700)], "eot")
    g = torch.Generator().manual_seed(77)
    audios = [(0.05 * torch.randn(int(sec * 16000), generator=g)).float() for sec in (85.0, 70.0, 20.0)]
    wins = [[a, b, c, c], [a, c, c], [c, c]]
    opts = dict(language="en", fp16=False, sample_len=limit, no_speech_threshold=1e-12, logprob_threshold=-4.0)
    singles, recorded = [], []
    for audio, w_ in zip(audios, wins):
        sc = set_script(Script(w_))
        try:
            singles.append(wt.transcribe(model, audio, **opts))
        finally:
            set_script(None)
        recorded.append(sc.record)
    assert len(recorded[0]) >= 3                                      # (A, the skipped B, C)
    scripts = [Script(r) for r in recorded]

    def on_group(idx):
        for i in idx:
            scripts[i].begin_window()
        set_row_scripts([scripts[i] for i in idx])
    streams.ON_GROUP_DECODE = on_group
    before = dict(streams.FALLBACK_READS)
    try:
        batch = wt.transcribe_batch(model, audios, max_streams=3, **opts)
    finally:
        streams.ON_GROUP_DECODE = None
        set_row_scripts(None)
    assert streams.FALLBACK_READS["from_the_last_row_kept_whole"] > before["from_the_last_row_kept_whole"]   # the path under test ran
    n_words = 0
    for b_, s_ in zip(batch, singles):
        vb, vs = (json.loads(json.dumps(G.public_view(x), default=float)) for x in (b_, s_))
        compare(vb, vs, time_tol=0.0, conf_tol=2e-5, logprob_tol=1e-5)
        n_words += sum(len(x["words"]) for x in vb["segments"])
    assert n_words > 0


def test_degenerate_batches(monkeypatch):
    """One ring block for several recordings (strictly one after the other through the B-stream driver), an EMPTY
    recording and a very short one among real ones: every recording still gets the dictionary transcribe() gives it."""
    import whisper_double as W
    from whisper_double.decoding import Script, set_row_scripts, set_script
    W.install()
    import whisper_timestamped as wt
    from whisper_timestamped import streams
    cpu_kernel_standin.install(monkeypatch)
    install_streams_standin(monkeypatch)
    # (a) max_streams = 1
    cases = same_model_cases()[:3]
    for raw, case in zip(run_batch(cases, max_streams=1), cases):
        compare(rounded(raw), case["expected"], time_tol=0.0, conf_tol=0.0, logprob_tol=1e-5)
    assert streams.LAST_RUN["ring_blocks"] == 1 and streams.LAST_RUN["admissions"] == 3
    # (b) an empty recording and a 0.3 s one between two real ones (scripts: whatever the model says, greedy)
    case = same_model_cases()[0]
    model, audio, _ = G.build_case(case, device="cpu")
    audios = [audio, torch.zeros(0), audio[:4800], audio]
    singles = []
    for a in audios:
        set_script(None)
        singles.append(wt.transcribe(model, a, language="en", fp16=False, sample_len=12))
    streams.ON_GROUP_DECODE = None
    set_row_scripts(None)
    batch = wt.transcribe_batch(model, audios, max_streams=3, language="en", fp16=False, sample_len=12)
    assert batch[1]["segments"] == [] and batch[1]["text"] == ""
    for b, s_ in zip(batch, singles):
        vb, vs = (json.loads(json.dumps(G.public_view(x), default=float)) for x in (b, s_))
        compare(vb, vs, time_tol=0.0, conf_tol=1e-3 + 1e-9, logprob_tol=1e-4)


def check_logits_view(device, write_digest=None):
    """_LogitsView over StreamRings' digests against the full rows it no longer keeps: the sampled token's log-probability,
    the argmax, the log-probability of the argmax / <|endoftext|> / <|notimestamps|> / any timestamp token, the argmax over
    later timestamps (T.py:535) -- and a loud refusal for anything else.  Shared by the CPU test (stand-in digest) and the
    GPU test (wt_logprob_digest_streams)."""
    import numpy as np
    import whisper_double as W
    from whisper_timestamped import _lib, streams
    W.install()
    model = W.build_model("tiny", seed=0, device=device)
    tk = W.tokenizer.get_tokenizer(True, language="en", task="transcribe")
    heads = torch.tensor([[2, 1], [3, 0], [3, 4]])
    n_streams, g, n_steps = 5, 3, 7
    rings = streams.StreamRings(model, heads, [0, 1, 2, 3], n_streams, torch.float32, sample_len=20, tokenizer=tk)
    assert rings.capacity == 21 and rings.digest.shape == (n_streams, 21, 8)
    V, ts0 = model.dims.n_vocab, tk.timestamp_begin
    gen = torch.Generator().manual_seed(3)
    blocks = [4, 0, 2]
    ring_index = torch.tensor(blocks, dtype=torch.int32, device=device)
    full = torch.randn((n_steps, g, V), generator=gen) * 3
    full[:, :, tk.no_timestamps] = float("-inf")
    full[2, 1, 700] = full[2, 1, 9000] = full[2, 1].max() + 1.0             # a tie: the FIRST index wins, as torch.argmax
    sampled = torch.randint(0, V, (n_steps, g), generator=gen)
    sampled[3, 0] = ts0 + 40
    before = rings.digest.clone()
    for step in range(n_steps):
        # the prompt call's rows come out of a (g, n_q, V) tensor (row stride n_q * V); tokens as a strided int64 column
        outs = torch.zeros((g, 3, V))
        outs[:, -1] = full[step]
        toks = torch.zeros((g, 4), dtype=torch.int64)
        toks[:, -1] = sampled[step]
        rings.write_digest(outs.to(device)[:, -1], toks.to(device)[:, -1], ring_index, step)
    untouched = [b for b in range(n_streams) if b not in blocks]
    assert torch.equal(rings.digest[untouched], before[untouched]) and float(rings.slice[untouched].abs().sum()) == 0.0
    assert float(rings.digest[blocks, n_steps:].abs().sum()) == 0.0
    host = rings.digest[ring_index.long(), :n_steps].cpu().numpy()
    ref = torch.log_softmax(full.float(), dim=-1)                              # (n_steps, g, V)
    for j, block in enumerate(blocks):
        view = streams._LogitsView(rings, block)
        view.window(host[j], sampled[:, j].tolist())
        for _ in range(n_steps):
            view.append(None)
        want = ref[torch.arange(n_steps), j, sampled[:, j]]
        got = view.gather(sampled[:, j].tolist())
        assert float((got - want).abs().max()) <= 2e-5
        for step in range(n_steps):
            assert view.argmax(step) == int(full[step, j].argmax()), (j, step)
        assert view.argmax(-1) == int(full[n_steps - 1, j].argmax())
        # the tokens the fallbacks name: the row's argmax, eot, a timestamp that was not sampled
        asked = sampled[:, j].tolist()
        asked[1] = int(full[1, j].argmax())
        asked[4] = int(tk.eot)
        asked[5] = ts0 + 123
        asked[6] = ts0 + 1500
        want = ref[torch.arange(n_steps), j, torch.tensor(asked)]
        got = view.gather(asked)
        assert float((got - want).abs().max()) <= 2e-5, (got, want)
        lo = ts0 + 41
        assert view.argmax(3, lo=lo) == int(full[3, j, lo:].argmax()) + lo
        assert view.argmax(-2, lo=ts0 + 1) == int(full[n_steps - 2, j, ts0 + 1:].argmax()) + ts0 + 1
        asked[2] = 1234 if 1234 not in (int(sampled[2, j]), int(full[2, j].argmax())) else 1235
        with pytest.raises(_lib.WtError, match="not kept"):
            view.gather(asked)
        # the loop's LAST call is kept whole (StreamRings.keep_last_rows): an arbitrary text token can be asked of it -- the
        # fallback token a stuck decoder's window takes from the NEXT window's prompt (T.py:498-503)
        asked[2] = int(sampled[2, j])
        asked[n_steps - 1] = 4321 if 4321 != int(sampled[n_steps - 1, j]) else 4322
        with pytest.raises(_lib.WtError, match="not kept"):
            view.gather(asked)                                           # (not before the row has been declared the last one)
        rings.keep_last_rows(full[n_steps - 1].to(device), ring_index.long())
        view.window(host[j], sampled[:, j].tolist(), full_row=n_steps - 1)
        got = view.gather(asked)
        want = ref[torch.arange(n_steps), j, torch.tensor(asked)]
        assert float((got - want).abs().max()) <= 2e-5, (got, want)
    assert int(host[1, 2, 3:4].view(np.int32)[0]) == 700
    return rings, full, sampled, host


def test_logits_view_serves_every_token_the_state_machine_can_ask_for(monkeypatch):
    cpu_kernel_standin.install(monkeypatch)
    install_streams_standin(monkeypatch)
    check_logits_view("cpu")


def test_vectorized_timestamp_rules_are_kept_only_when_they_reproduce_the_backend(monkeypatch):
    """An older backend (no 'timestamps never decrease' rule): the probe notices, the driver keeps the backend's filter."""
    import whisper_double as W
    from whisper_timestamped import streams
    W.install()
    tk = W.tokenizer.get_tokenizer(True, language="en", task="transcribe")
    streams._RULES_PROBED.clear()
    rule = W.decoding.ApplyTimestampRules(tk, 4, 50)
    task = type("T", (), {"logit_filters": [rule, W.decoding.SuppressTokens([1, 2])]})()
    out = streams.vectorize_filters(task).logit_filters
    assert isinstance(out[0], streams.BatchedTimestampRules) and isinstance(out[1], streams.BatchedSuppressTokens)

    class ApplyTimestampRules(W.decoding.ApplyTimestampRules):         # the 2022 rule set: pairs + first position + mass
        def apply(self, logits, tokens):
            import torch.nn.functional as F
            tk_, ts0 = self.tokenizer, self.tokenizer.timestamp_begin
            logits[:, tk_.no_timestamps] = -float("inf")
            for k in range(tokens.shape[0]):
                seq = tokens[k, self.sample_begin:].tolist()
                last = len(seq) >= 1 and seq[-1] >= ts0
                penult = len(seq) < 2 or seq[-2] >= ts0
                if last:
                    if penult:
                        logits[k, ts0:] = -float("inf")
                    else:
                        logits[k, :tk_.eot] = -float("inf")
            if tokens.shape[1] == self.sample_begin:
                logits[:, :ts0] = -float("inf")
                if self.max_initial_timestamp_index is not None:
                    logits[:, ts0 + self.max_initial_timestamp_index + 1:] = -float("inf")
            lp = F.log_softmax(logits.float(), dim=-1)
            for k in range(tokens.shape[0]):
                if lp[k, ts0:].logsumexp(dim=-1) > lp[k, :ts0].max():
                    logits[k, :ts0] = -float("inf")
    old = ApplyTimestampRules(tk, 4, 50)
    task = type("T", (), {"logit_filters": [old]})()
    assert streams.vectorize_filters(task).logit_filters[0] is old
    streams._RULES_PROBED.clear()


def test_fallback_paths_of_transcribe_batch_reuse_the_model_the_plan_loaded(monkeypatch):
    """with `model` given as a NAME, transcribe_batch's one-by-one fallbacks (calls the B-stream path cannot take;
    a backend without DecodingTask._main_loop) must decode every recording with the model _plan has already loaded -- not
    load it again per recording."""
    import whisper_double as W
    W.install()
    import importlib
    import whisper_timestamped as wt
    T = importlib.import_module("whisper_timestamped.transcribe")          # (the package re-exports the FUNCTION under this name)
    cpu_kernel_standin.install(monkeypatch)
    install_streams_standin(monkeypatch)
    model = W.build_model("tiny", seed=0, device="cpu")
    loads = []
    monkeypatch.setattr(T, "load_model", lambda name, *a, **k: (loads.append(name), model)[1])
    g = torch.Generator().manual_seed(1)
    audios = [(0.05 * torch.randn(16000 * 3, generator=g)).float() for _ in range(3)]
    # (a) beam search: not batchable -> one recording at a time
    out = wt.transcribe_batch("tiny", audios, language="en", fp16=False, beam_size=2, sample_len=6)
    assert len(out) == 3 and loads == ["tiny"], loads
    # (b) a backend that lacks the decoder loop as a method
    loads.clear()
    from whisper_timestamped import streams
    monkeypatch.setattr(streams, "backend_missing", lambda: ["DecodingTask._main_loop"])
    out = wt.transcribe_batch("tiny", audios, language="en", fp16=False, sample_len=6)
    assert len(out) == 3 and loads == ["tiny"], loads
