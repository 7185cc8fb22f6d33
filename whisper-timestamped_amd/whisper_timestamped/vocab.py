"""Whisper tokenizer from a ``.tiktoken`` vocabulary file -- without the ``tiktoken`` package.

The reference obtains its tokenizer from the ASR backend (``whisper.tokenizer.get_tokenizer``,
/root/reference/whisper_timestamped/transcribe.py:1406-1426) and only ever uses this surface of it on the
word-alignment path: ``decode`` / ``decode_with_timestamps`` (token -> word splitting, transcribe.py:1815-1868),
``timestamp_begin`` / ``eot`` / ``sot`` / ``sot_sequence`` / ``to_language_token`` / ``all_language_*`` (hook state
machine, transcribe.py:419-420, 811-832, 862-867), ``encode`` (initial prompts, the punctuation option of the
confidence path).  openai-whisper ships the vocabulary as two text files, ``assets/multilingual.tiktoken`` and
``assets/gpt2.tiktoken``: one ``base64(token bytes) rank`` pair per line.  This module reads such a file and builds an
object with that surface, so that

  * the reference's end-to-end goldens (tests/expected/*.words.json) can be replayed wherever the vocabulary file and
    trained weights exist, even if ``tiktoken`` (a compiled extension) does not (tools/replay_reference_goldens.py);
  * a backend other than openai-whisper (a model exported with its vocabulary next to it) gets the same word splitting.

The special-token layout is openai-whisper's: ``<|endoftext|>`` = number of ranks, then
``<|startoftranscript|>``, the language tokens (99, or 100 for large-v3 / turbo), translate, transcribe, startoflm,
startofprev, nospeech, notimestamps and the 1501 timestamps ``<|0.00|>`` ... ``<|30.00|>``.

Byte-pair encoding is the textbook algorithm over the ranks (split with GPT-2's pattern, then repeatedly merge the
adjacent pair with the lowest rank): identical to tiktoken's result by construction, only slower -- encoding is off the
hot path (prompts and a handful of symbol strings).
"""
from __future__ import annotations

import base64
import os
from functools import cached_property, lru_cache

LANGUAGES = {
    "en": "english", "zh": "chinese", "de": "german", "es": "spanish", "ru": "russian", "ko": "korean", "fr": "french",
    "ja": "japanese", "pt": "portuguese", "tr": "turkish", "pl": "polish", "ca": "catalan", "nl": "dutch", "ar": "arabic",
    "sv": "swedish", "it": "italian", "id": "indonesian", "hi": "hindi", "fi": "finnish", "vi": "vietnamese",
    "he": "hebrew", "uk": "ukrainian", "el": "greek", "ms": "malay", "cs": "czech", "ro": "romanian", "da": "danish",
    "hu": "hungarian", "ta": "tamil", "no": "norwegian", "th": "thai", "ur": "urdu", "hr": "croatian", "bg": "bulgarian",
    "lt": "lithuanian", "la": "latin", "mi": "maori", "ml": "malayalam", "cy": "welsh", "sk": "slovak", "te": "telugu",
    "fa": "persian", "lv": "latvian", "bn": "bengali", "sr": "serbian", "az": "azerbaijani", "sl": "slovenian",
    "kn": "kannada", "et": "estonian", "mk": "macedonian", "br": "breton", "eu": "basque", "is": "icelandic",
    "hy": "armenian", "ne": "nepali", "mn": "mongolian", "bs": "bosnian", "kk": "kazakh", "sq": "albanian",
    "sw": "swahili", "gl": "galician", "mr": "marathi", "pa": "punjabi", "si": "sinhala", "km": "khmer", "sn": "shona",
    "yo": "yoruba", "so": "somali", "af": "afrikaans", "oc": "occitan", "ka": "georgian", "be": "belarusian",
    "tg": "tajik", "sd": "sindhi", "gu": "gujarati", "am": "amharic", "yi": "yiddish", "lo": "lao", "uz": "uzbek",
    "fo": "faroese", "ht": "haitian creole", "ps": "pashto", "tk": "turkmen", "nn": "nynorsk", "mt": "maltese",
    "sa": "sanskrit", "lb": "luxembourgish", "my": "myanmar", "bo": "tibetan", "tl": "tagalog", "mg": "malagasy",
    "as": "assamese", "tt": "tatar", "haw": "hawaiian", "ln": "lingala", "ha": "hausa", "ba": "bashkir",
    "jw": "javanese", "su": "sundanese", "yue": "cantonese",
}
TO_LANGUAGE_CODE = {
    **{name: code for code, name in LANGUAGES.items()},
    "burmese": "my", "valencian": "ca", "flemish": "nl", "haitian": "ht", "letzeburgesch": "lb", "pushto": "ps",
    "panjabi": "pa", "moldavian": "ro", "moldovan": "ro", "sinhalese": "si", "castilian": "es", "mandarin": "zh",
}

# GPT-2's pre-tokenisation pattern (what openai-whisper hands to tiktoken.Encoding as pat_str)
GPT2_PATTERN = r"""'s|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"""


def load_ranks(path: str) -> dict:
    """``base64(token) rank`` lines -> {token bytes: rank} (the format of whisper/assets/*.tiktoken)."""
    ranks = {}
    with open(path, "rb") as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            token, rank = line.split()
            ranks[base64.b64decode(token)] = int(rank)
    if sorted(ranks.values()) != list(range(len(ranks))):
        raise ValueError(f"{path}: ranks are not a permutation of 0..{len(ranks) - 1}")
    return ranks


def special_token_names(num_languages: int):
    names = ["<|endoftext|>", "<|startoftranscript|>"]
    names += [f"<|{code}|>" for code in list(LANGUAGES)[:num_languages]]
    names += ["<|translate|>", "<|transcribe|>", "<|startoflm|>", "<|startofprev|>", "<|nospeech|>", "<|notimestamps|>"]
    names += [f"<|{i * 0.02:.2f}|>" for i in range(1501)]
    return names


def _splitter():
    try:
        import regex
        return regex.compile(GPT2_PATTERN).findall
    except ImportError:                                   # pragma: no cover - `regex` ships with this image
        import re
        # \p{L} / \p{N} approximated by str.isalpha / isnumeric classes of `re`: letters = [^\W\d_], digits = \d
        pat = r"""'s|'t|'re|'ve|'m|'ll|'d| ?[^\W\d_]+| ?\d+| ?(?:[^\s\w]|_)+|\s+(?!\S)|\s+"""
        return re.compile(pat).findall


class Tokenizer:
    """The attributes and methods of ``whisper.tokenizer.Tokenizer`` that whisper-timestamped and openai-whisper's decoding
    loop read, over ranks loaded from a ``.tiktoken`` file."""

    def __init__(self, ranks: dict, num_languages: int = 99, language=None, task=None, name: str = "custom"):
        self.name = name
        self.num_languages = num_languages
        self.language, self.task = language, task
        self._ranks = ranks
        self._piece_of = [None] * len(ranks)
        for piece, rank in ranks.items():
            self._piece_of[rank] = piece
        base = len(ranks)
        names = special_token_names(num_languages)
        self.special_tokens = {n: base + i for i, n in enumerate(names)}
        self._special_of = {v: k for k, v in self.special_tokens.items()}
        self.n_vocab = base + len(names)
        self._split = _splitter()
        sot_sequence = [self.sot]
        if language is not None:
            sot_sequence.append(self.to_language_token(language))
        if task is not None:
            sot_sequence.append(self.transcribe if task == "transcribe" else self.translate)
        self.sot_sequence = tuple(sot_sequence)

    # ---- special ids --------------------------------------------------------------------------------------------
    eot = property(lambda self: self.special_tokens["<|endoftext|>"])
    sot = property(lambda self: self.special_tokens["<|startoftranscript|>"])
    transcribe = property(lambda self: self.special_tokens["<|transcribe|>"])
    translate = property(lambda self: self.special_tokens["<|translate|>"])
    sot_lm = property(lambda self: self.special_tokens["<|startoflm|>"])
    sot_prev = property(lambda self: self.special_tokens["<|startofprev|>"])
    no_speech = property(lambda self: self.special_tokens["<|nospeech|>"])
    no_timestamps = property(lambda self: self.special_tokens["<|notimestamps|>"])
    timestamp_begin = property(lambda self: self.special_tokens["<|0.00|>"])

    @property
    def language_token(self) -> int:
        if self.language is None:
            raise ValueError("This tokenizer does not have language token configured")
        return self.to_language_token(self.language)

    def to_language_token(self, language) -> int:
        tok = self.special_tokens.get(f"<|{language}|>")
        if tok is None:
            raise KeyError(f"Language {language} not found in tokenizer.")
        return tok

    @cached_property
    def all_language_tokens(self):
        codes = set(LANGUAGES)
        return tuple(tid for name, tid in self.special_tokens.items() if name.strip("<|>") in codes)[: self.num_languages]

    @cached_property
    def all_language_codes(self):
        return tuple(self._special_of[t].strip("<|>") for t in self.all_language_tokens)

    @cached_property
    def sot_sequence_including_notimestamps(self):
        return tuple(list(self.sot_sequence) + [self.no_timestamps])

    @cached_property
    def non_speech_tokens(self):
        """openai-whisper's list of symbol tokens to suppress (speaker tags, annotations, music notes): ids of the
        symbols that encode to ONE token, with and without a leading space; `` -`` and `` '`` always."""
        symbols = list('"#()*+/:;<=>@[\\]^_`{|}~「」『』')
        symbols += "<< >> <<< >>> -- --- -( -[ (' (\" (( )) ((( ))) [[ ]] {{ }} ♪♪ ♪♪♪".split()
        miscellaneous = set("♩♪♫♬♭♮♯")
        result = {self.encode(" -")[0], self.encode(" '")[0]}
        for symbol in symbols + list(miscellaneous):
            for tokens in (self.encode(symbol), self.encode(" " + symbol)):
                if len(tokens) == 1 or symbol in miscellaneous:
                    result.add(tokens[0])
        return tuple(sorted(result))

    # ---- text <-> ids -----------------------------------------------------------------------------------------------
    @lru_cache(maxsize=65536)
    def _bpe(self, piece: bytes):
        parts = [bytes([b]) for b in piece]
        ranks = self._ranks
        while len(parts) > 1:
            best, best_rank = -1, None
            for i in range(len(parts) - 1):
                r = ranks.get(parts[i] + parts[i + 1])
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = i, r
            if best < 0:
                break
            parts[best:best + 2] = [parts[best] + parts[best + 1]]
        return tuple(ranks[p] for p in parts)

    def encode(self, text: str, **kwargs):
        out = []
        for piece in self._split(text):
            data = piece.encode("utf-8")
            r = self._ranks.get(data)
            if r is not None:
                out.append(r)
            else:
                out.extend(self._bpe(data))
        return out

    def _bytes_of(self, t: int) -> bytes:
        t = int(t)
        if t < len(self._piece_of):
            return self._piece_of[t]
        name = self._special_of.get(t)
        if name is None:
            raise KeyError(f"token id {t} outside the vocabulary ({self.n_vocab})")
        return name.encode()

    def decode(self, token_ids, **kwargs) -> str:
        """Text of the ids below ``timestamp_begin`` (timestamp tokens are dropped, as whisper's ``decode`` does)."""
        ts0 = self.timestamp_begin
        return b"".join(self._bytes_of(t) for t in token_ids if int(t) < ts0).decode("utf-8", errors="replace")

    def decode_with_timestamps(self, token_ids, **kwargs) -> str:
        """Timestamp tokens are annotated, e.g. "<|1.08|>"."""
        return b"".join(self._bytes_of(t) for t in token_ids).decode("utf-8", errors="replace")


def find_vocab_file(multilingual: bool):
    """Where a ``.tiktoken`` file of the right kind lives: $WT_TOKENIZER_VOCAB (a file, or a directory holding
    multilingual.tiktoken / gpt2.tiktoken), else the assets directory of an installed openai-whisper."""
    fname = "multilingual.tiktoken" if multilingual else "gpt2.tiktoken"
    env = os.environ.get("WT_TOKENIZER_VOCAB")
    if env:
        path = os.path.join(env, fname) if os.path.isdir(env) else env
        if os.path.isfile(path):
            return path
    try:
        import importlib.util
        spec = importlib.util.find_spec("whisper")
        if spec and spec.submodule_search_locations:
            for loc in spec.submodule_search_locations:
                path = os.path.join(loc, "assets", fname)
                if os.path.isfile(path):
                    return path
    except (ImportError, ValueError):
        pass
    return None


@lru_cache(maxsize=None)
def _ranks_of(path: str):
    return load_ranks(path)


def get_tokenizer(multilingual: bool, *, num_languages: int = 99, language=None, task=None, vocab_path=None) -> Tokenizer:
    """Same call as ``whisper.tokenizer.get_tokenizer`` (+ ``vocab_path``)."""
    if language is not None:
        language = language.lower()
        if language not in LANGUAGES:
            if language in TO_LANGUAGE_CODE:
                language = TO_LANGUAGE_CODE[language]
            else:
                raise ValueError(f"Unsupported language: {language}")
    if multilingual:
        language, task = language or "en", task or "transcribe"
    else:
        language = task = None
    path = vocab_path or find_vocab_file(multilingual)
    if path is None:
        raise FileNotFoundError("no .tiktoken vocabulary: pass vocab_path=, set WT_TOKENIZER_VOCAB, or install openai-whisper "
                                "(whisper/assets/multilingual.tiktoken, gpt2.tiktoken)")
    return Tokenizer(_ranks_of(os.path.abspath(path)), num_languages=num_languages, language=language, task=task,
                     name="multilingual" if multilingual else "gpt2")
