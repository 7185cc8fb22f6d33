"""whisper.tokenizer stand-in.

The real vocabulary (tiktoken ranks) is not available offline.  What IS known
and reproduced: the special-token layout, the GPT-2 byte
table (ids 0..255 are single bytes in GPT-2's printable-first order: this is
what makes id 220 == " ", 11 == ",", 13 == "." as in the reference's
known-answer test), and the pieces listed by that test
(/root/reference/tests/test_transcribe.py:722-902).  Every other id below
``eot`` gets a deterministic synthetic piece (tests/synth.py).
"""
from functools import cached_property, lru_cache

import synth

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
assert len(LANGUAGES) == 100

TO_LANGUAGE_CODE = {
    **{name: code for code, name in LANGUAGES.items()},
    "burmese": "my", "valencian": "ca", "flemish": "nl", "haitian": "ht", "letzeburgesch": "lb", "pushto": "ps",
    "panjabi": "pa", "moldavian": "ro", "moldovan": "ro", "sinhalese": "si", "castilian": "es", "mandarin": "zh",
}


def _gpt2_byte_table():
    """id (0..255) -> byte value, GPT-2 order: printable latin-1 first, then the rest."""
    printable = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    rest = [b for b in range(256) if b not in printable]
    return printable + rest


_BYTE_OF_ID = _gpt2_byte_table()
_ID_OF_BYTE = {b: i for i, b in enumerate(_BYTE_OF_ID)}


class Tokenizer:
    def __init__(self, multilingual: bool, num_languages: int = 99, language=None, task=None):
        self.multilingual = multilingual
        self.num_languages = num_languages
        self.language = language
        self.task = task
        base = 50257 if multilingual else 50256
        names = ["<|endoftext|>", "<|startoftranscript|>"]
        names += [f"<|{code}|>" for code in list(LANGUAGES)[:num_languages]]
        names += ["<|translate|>", "<|transcribe|>", "<|startoflm|>", "<|startofprev|>", "<|nospeech|>", "<|notimestamps|>"]
        names += [f"<|{i * 0.02:.2f}|>" for i in range(1501)]
        self.special_tokens = {name: base + i for i, name in enumerate(names)}
        self._special_names = {v: k for k, v in self.special_tokens.items()}
        self.n_vocab = base + len(names)
        self.vocab = dict(synth.KAT_VOCAB_BYTES)
        if not multilingual:
            self.vocab.update(synth.KAT_VOCAB_EN)
        sot_sequence = [self.sot]
        if language is not None:
            sot_sequence.append(self.sot + 1 + list(LANGUAGES).index(language))
        if task is not None:
            sot_sequence.append(self.transcribe if task == "transcribe" else self.translate)
        self.sot_sequence = tuple(sot_sequence)

    # ---- special ids -----------------------------------------------------
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
        out = [tid for name, tid in self.special_tokens.items() if name.strip("<|>") in codes]
        return tuple(out)[: self.num_languages]

    @cached_property
    def all_language_codes(self):
        return tuple(self.decode([t]).strip("<|>") for t in self.all_language_tokens)

    @cached_property
    def sot_sequence_including_notimestamps(self):
        return tuple(list(self.sot_sequence) + [self.no_timestamps])

    @cached_property
    def non_speech_tokens(self):
        """Ids of symbol-only pieces (the real list is derived the same way from a symbol table)."""
        symbols = list('"#()*+/:;<=>@[\\]^_`{|}~「」『』') + ["<<", ">>", "<<<", ">>>", "--", "---", "-(", "-[", "('", '("',
                                                             "((", "))", "(((", ")))", "[[", "]]", "{{", "}}", "♪♪", "♪♪♪"]
        result = {self.encode(" -")[0], self.encode(" '")[0]}
        for s in symbols:
            for toks in (self.encode(s), self.encode(" " + s)):
                if len(toks) == 1:
                    result.add(toks[0])
        return tuple(sorted(result))

    # ---- text <-> ids -------------------------------------------------------
    def _piece(self, t: int) -> bytes:
        t = int(t)
        if t in self._special_names:
            return self._special_names[t].encode()
        if t < 256:
            return bytes([_BYTE_OF_ID[t]])
        v = self.vocab.get(t)
        return v if v is not None else synth._synthetic_piece(t)

    def encode(self, text, **kwargs):
        data = text.encode("utf-8")
        inv = self._inverse_vocab()
        out, i = [], 0
        while i < len(data):          # greedy longest match over the known pieces, else single bytes
            for n in range(min(8, len(data) - i), 1, -1):
                tok = inv.get(data[i:i + n])
                if tok is not None:
                    out.append(tok)
                    i += n
                    break
            else:
                out.append(_ID_OF_BYTE[data[i]])
                i += 1
        return out

    @lru_cache(maxsize=None)
    def _inverse_vocab(self):
        return {v: k for k, v in sorted(self.vocab.items(), reverse=True)}

    def decode(self, token_ids, **kwargs) -> str:
        token_ids = [int(t) for t in token_ids if int(t) < self.timestamp_begin]
        return b"".join(self._piece(t) for t in token_ids).decode("utf-8", errors="replace")

    def decode_with_timestamps(self, token_ids, **kwargs) -> str:
        return b"".join(self._piece(t) for t in token_ids).decode("utf-8", errors="replace")


@lru_cache(maxsize=None)
def get_tokenizer(multilingual: bool, *, num_languages: int = 99, language=None, task=None) -> Tokenizer:
    if language is not None:
        language = language.lower()
        if language not in LANGUAGES:
            if language in TO_LANGUAGE_CODE:
                language = TO_LANGUAGE_CODE[language]
            else:
                raise ValueError(f"Unsupported language: {language}")
    if multilingual:
        language = language or "en"
        task = task or "transcribe"
    else:
        language = None
        task = None
    return Tokenizer(multilingual, num_languages=num_languages, language=language, task=task)
