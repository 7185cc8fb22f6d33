"""whisper_timestamped/vocab.py: the package's own loader of ``.tiktoken`` vocabulary files.

The real vocabulary files are not in this image, so the loader is held against
  * a vocabulary TRAINED here (byte-pair merges learnt on a small corpus, written in the .tiktoken format), encoded by
    an independent formulation of BPE (GPT-2's original "merge every occurrence of the best-ranked pair" loop);
  * openai-whisper's special-token layout, through the ids the reference's own tests quote
    (/root/reference/tests/test_transcribe.py:733-738, 893-900: 50364 = "<|0.00|>", 50714 = "<|7.00|>" multilingual;
    eot 50256 / timestamp_begin 50363 English-only) and SURVEY.md Appendix C (large-v3: +1 after the languages);
  * the real thing wherever it exists: with openai-whisper + tiktoken installed the last test compares both on text.
"""
import base64
import collections
import os

import pytest

from whisper_timestamped import vocab as TK

CORPUS = ("the quick brown fox jumps over the lazy dog. " * 3 + "Bonjour, vous allez bien ? Oui, très bien, merci ! "
          "Let's go with it again! I'm sure they've said: \"it's 12:30\" -- isn't it? 日本語のテキスト、そして한국어 텍스트. "
          "whisper-timestamped aligns words with DTW over cross-attention weights; numbers 1234 567 89. " * 2)


def gpt2_byte_order():
    printable = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    return printable + [b for b in range(256) if b not in printable]


def train_ranks(n_merges, pad_to=None):
    """256 byte tokens in GPT-2's order, then `n_merges` merges learnt greedily (most frequent adjacent pair) on CORPUS
    split by the tokenizer's own pattern; optionally padded with unique unused 5-byte tokens up to `pad_to` ranks."""
    ranks = {bytes([b]): i for i, b in enumerate(gpt2_byte_order())}
    words = [[bytes([b]) for b in w.encode("utf-8")] for w in TK._splitter()(CORPUS)]
    for _ in range(n_merges):
        pairs = collections.Counter()
        for w in words:
            for a, b in zip(w, w[1:]):
                pairs[(a, b)] += 1
        if not pairs:
            break
        (a, b), _n = max(pairs.items(), key=lambda kv: (kv[1], kv[0]))
        ranks[a + b] = len(ranks)
        for w in words:
            i = 0
            while i < len(w) - 1:
                if w[i] == a and w[i + 1] == b:
                    w[i:i + 2] = [a + b]
                else:
                    i += 1
    k = 0
    while pad_to is not None and len(ranks) < pad_to:
        tok = b"\xff\xfe" + k.to_bytes(3, "big")         # never produced by UTF-8 text
        k += 1
        if tok not in ranks:
            ranks[tok] = len(ranks)
    return ranks


def write_vocab(path, ranks):
    with open(path, "wb") as f:
        for tok, r in sorted(ranks.items(), key=lambda kv: kv[1]):
            f.write(base64.b64encode(tok) + b" " + str(r).encode() + b"\n")


def gpt2_bpe(piece: bytes, ranks):
    """GPT-2's own loop: find the best-ranked bigram, merge ALL its occurrences, repeat."""
    word = [bytes([b]) for b in piece]
    while len(word) > 1:
        pairs = {(a, b) for a, b in zip(word, word[1:])}
        best = min(pairs, key=lambda p: ranks.get(p[0] + p[1], float("inf")))
        if best[0] + best[1] not in ranks:
            break
        a, b = best
        out, i = [], 0
        while i < len(word):
            if i < len(word) - 1 and word[i] == a and word[i + 1] == b:
                out.append(a + b)
                i += 2
            else:
                out.append(word[i])
                i += 1
        word = out
    return [ranks[w] for w in word]


@pytest.fixture(scope="module")
def vocab_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("vocab")
    write_vocab(d / "multilingual.tiktoken", train_ranks(400, pad_to=50257))
    write_vocab(d / "gpt2.tiktoken", train_ranks(300, pad_to=50256))
    return str(d)


def test_file_round_trip_and_rank_check(tmp_path):
    ranks = train_ranks(50)
    write_vocab(tmp_path / "v.tiktoken", ranks)
    assert TK.load_ranks(str(tmp_path / "v.tiktoken")) == ranks
    broken = dict(ranks)
    broken[b"zz-gap"] = len(ranks) + 5
    write_vocab(tmp_path / "bad.tiktoken", broken)
    with pytest.raises(ValueError):
        TK.load_ranks(str(tmp_path / "bad.tiktoken"))


def test_special_token_layout_is_openai_whispers(vocab_dir):
    multi = TK.get_tokenizer(True, language="fr", task="transcribe", vocab_path=os.path.join(vocab_dir, "multilingual.tiktoken"))
    assert (multi.eot, multi.sot, multi.translate, multi.transcribe) == (50257, 50258, 50358, 50359)
    assert (multi.sot_lm, multi.sot_prev, multi.no_speech, multi.no_timestamps, multi.timestamp_begin) == \
        (50360, 50361, 50362, 50363, 50364)
    assert multi.decode_with_timestamps([50364]) == "<|0.00|>" and multi.decode_with_timestamps([50714]) == "<|7.00|>"
    assert multi.n_vocab == 51865 and len(multi.all_language_tokens) == 99
    assert multi.sot_sequence == (50258, multi.to_language_token("fr"), 50359) and multi.to_language_token("en") == 50259
    assert multi.all_language_codes[:3] == ("en", "zh", "de") and multi.all_language_codes[-1] == "su"
    en = TK.get_tokenizer(False, language="en", task="transcribe", vocab_path=os.path.join(vocab_dir, "gpt2.tiktoken"))
    assert (en.eot, en.sot, en.timestamp_begin, en.n_vocab) == (50256, 50257, 50363, 51864)
    assert en.sot_sequence == (50257,) and en.language is None
    v3 = TK.get_tokenizer(True, num_languages=100, vocab_path=os.path.join(vocab_dir, "multilingual.tiktoken"))
    assert (v3.translate, v3.timestamp_begin, v3.n_vocab) == (50359, 50365, 51866)
    assert v3.all_language_codes[-1] == "yue" and v3.sot_sequence == (50258, 50259, 50360)
    with pytest.raises(KeyError):
        multi.to_language_token("xx")
    with pytest.raises(ValueError):
        TK.get_tokenizer(True, language="klingon", vocab_path=os.path.join(vocab_dir, "multilingual.tiktoken"))
    assert TK.get_tokenizer(True, language="French", vocab_path=os.path.join(vocab_dir, "multilingual.tiktoken")).language == "fr"


def test_encode_is_bpe_and_decode_inverts_it(vocab_dir):
    tok = TK.get_tokenizer(True, vocab_path=os.path.join(vocab_dir, "multilingual.tiktoken"))
    ranks = TK.load_ranks(os.path.join(vocab_dir, "multilingual.tiktoken"))
    texts = [CORPUS, " -", " '", "it's the quickest fox!?", "  two  spaces\nand a newline ", "naïve café — 東京 12345", "♪♪ [music]"]
    for text in texts:
        ids = tok.encode(text)
        want = [t for piece in TK._splitter()(text) for t in gpt2_bpe(piece.encode("utf-8"), ranks)]
        assert ids == want, text
        assert tok.decode(ids) == text
        assert all(t < tok.eot for t in ids)
    merged = [t for t in tok.encode(CORPUS) if t >= 256]
    assert len(merged) > 50                       # the learnt merges are really used
    # timestamps are dropped by decode() and shown by decode_with_timestamps(), as in whisper
    ids = [tok.timestamp_begin + 54] + tok.encode(" hello") + [tok.timestamp_begin + 100]
    assert tok.decode(ids) == " hello" and tok.decode_with_timestamps(ids) == "<|1.08|> hello<|2.00|>"
    assert tok.decode([tok.eot]) == "<|endoftext|>"
    # a piece cut in the middle of a UTF-8 sequence decodes with the replacement character (what the word splitter
    # of transcribe.py:1815-1842 relies on)
    cut = tok.encode("é")
    assert "�" in tok.decode([gpt2_byte_order().index("é".encode("utf-8")[0])]) or len(cut) == 1


def test_non_speech_tokens_are_single_symbol_tokens(vocab_dir):
    tok = TK.get_tokenizer(True, vocab_path=os.path.join(vocab_dir, "multilingual.tiktoken"))
    ns = tok.non_speech_tokens
    assert tok.encode(" -")[0] in ns and tok.encode(" '")[0] in ns
    assert tok.encode("(")[0] in ns and tok.encode("[")[0] in ns and tok.encode("♪")[0] in ns
    assert tok.encode("a")[0] not in ns and list(ns) == sorted(set(ns))


def test_backend_seam_uses_the_vocabulary_file(vocab_dir, monkeypatch):
    """backend.get_tokenizer (transcribe.py:1406-1426) with $WT_TOKENIZER_VOCAB / model.tokenizer_vocab."""
    from whisper_timestamped import backend

    class M:
        is_multilingual = True
        num_languages = 100
    monkeypatch.setenv("WT_TOKENIZER_VOCAB", vocab_dir)
    t = backend.get_tokenizer(M(), task="translate", language="de")
    assert isinstance(t, TK.Tokenizer) and t.timestamp_begin == 50365 and t.sot_sequence == (50258, t.to_language_token("de"), t.translate)
    monkeypatch.delenv("WT_TOKENIZER_VOCAB")
    m = M()
    m.is_multilingual, m.num_languages = False, 99
    m.tokenizer_vocab = os.path.join(vocab_dir, "gpt2.tiktoken")
    assert backend.get_tokenizer(m).eot == 50256


def test_word_splitting_runs_on_the_loaded_tokenizer(vocab_dir):
    """The consumer on the alignment path: split_tokens_on_spaces (transcribe.py:1845-1868) over ids of this tokenizer."""
    from whisper_timestamped.words import split_tokens_on_spaces
    tok = TK.get_tokenizer(True, vocab_path=os.path.join(vocab_dir, "multilingual.tiktoken"))
    ids = [tok.timestamp_begin] + tok.encode(" Let's go with it again!") + [tok.timestamp_begin + 120]
    words, pieces, ids_per_word = split_tokens_on_spaces(ids, tok)
    assert words == ["<|0.00|>", "Let's", "go", "with", "it", "again!", "<|2.40|>"]      # (texts stripped: transcribe.py:1860)
    assert pieces[1][0] == " Let" and pieces[-2][-1] == "!"
    assert [t for w in ids_per_word for t in w] == ids


def test_against_openai_whisper_where_installed():
    """On a box with openai-whisper (+ tiktoken): same ids and same text as the backend's own tokenizer."""
    pytest.importorskip("tiktoken")
    whisper_tk = pytest.importorskip("whisper.tokenizer")
    for multilingual, nl in ((True, 99), (True, 100), (False, 99)):
        path = TK.find_vocab_file(multilingual)
        assert path is not None
        ours = TK.get_tokenizer(multilingual, num_languages=nl, language="fr", task="transcribe", vocab_path=path)
        theirs = whisper_tk.get_tokenizer(multilingual, num_languages=nl, language="fr", task="transcribe")
        assert ours.sot_sequence == tuple(theirs.sot_sequence) and ours.timestamp_begin == theirs.timestamp_begin
        assert ours.eot == theirs.eot and ours.no_speech == theirs.no_speech
        assert tuple(ours.non_speech_tokens) == tuple(theirs.non_speech_tokens)
        for text in (CORPUS, " Mohoo! Let's go with it again!", "  x\n\ny ", "東京は12345です。"):
            assert ours.encode(text) == theirs.encode(text), text
            ids = theirs.encode(text)
            assert ours.decode(ids) == theirs.decode(ids)
            assert ours.decode_with_timestamps([ours.timestamp_begin + 7] + ids) == \
                theirs.decode_with_timestamps([theirs.timestamp_begin + 7] + ids)


def test_language_table_equals_transformers():
    """The ORDER of the language list decides every language token id.  Independent source: the table transformers ships
    for its own Whisper tokenizer (same 100 languages, same order, same aliases)."""
    hf = pytest.importorskip("transformers.models.whisper.tokenization_whisper")
    assert list(hf.LANGUAGES.items()) == list(TK.LANGUAGES.items())
    assert hf.TO_LANGUAGE_CODE == TK.TO_LANGUAGE_CODE
    names = TK.special_token_names(100)
    assert names[2] == "<|en|>" and names[101] == "<|yue|>" and names[102] == "<|translate|>" and len(names) == 2 + 100 + 6 + 1501
