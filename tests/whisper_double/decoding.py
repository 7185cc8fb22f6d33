"""whisper.decoding stand-in: options, logit filters, greedy / beam decoders, DecodingTask.

Extra (test-only) feature: a ``Script`` can be attached (``set_script``): the
sampler then returns the scripted token instead of the arg-max / beam choice,
while log-probabilities stay the model's own.
"""
from dataclasses import dataclass, field, replace
from typing import Iterable, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F
from torch.distributions import Categorical

from .audio import CHUNK_LENGTH
from .tokenizer import Tokenizer, get_tokenizer
from .utils import compression_ratio


# ---------------------------------------------------------------------------
class Script:
    """Per-window forced samples.  ``windows[k]`` = tokens the sampler must
    return during the k-th DecodingTask.run (must end with eot unless the
    window is meant to hit the sample_len limit).  ``record`` collects what was
    actually sampled (scripted or not), per window."""

    def __init__(self, windows=None):
        self.windows = [list(w) for w in windows] if windows is not None else None
        self.record = []
        self._k = -1
        self._i = 0

    def begin_window(self):
        self._k += 1
        self._i = 0
        self.record.append([])

    def pick(self, natural: int, logits_row=None, eot=None, text_best=None) -> int:
        tok = natural
        if self.windows is not None and self._k < len(self.windows) and self._i < len(self.windows[self._k]):
            tok = self.windows[self._k][self._i]
            if tok is None:       # "the most likely TEXT token": keeps the scripted structure, meaningful probabilities
                tok = int(logits_row[:eot].argmax()) if text_best is None else int(text_best)
        self._i += 1
        self.record[-1].append(int(tok))
        return int(tok)

    def final_tokens(self, natural, eot):
        """Beam search cannot be steered token by token: the scripted window replaces its RESULT."""
        if self.windows is not None and self._k < len(self.windows):
            forced = [t for t in self.windows[self._k] if t != eot]
            self.record[-1] = forced + [eot]
            return forced
        self.record[-1] = list(natural) + [eot]
        return natural


_SCRIPT: Optional[Script] = None
_ROW_SCRIPTS: Optional[list] = None      # B streams in one decoder loop: one Script (or None) per batch row


def set_script(script: Optional[Script]):
    global _SCRIPT
    _SCRIPT = script
    return script


def set_row_scripts(scripts: Optional[list]):
    """Batched decoding (whisper_timestamped.streams): row k of the next decoder loops is steered by scripts[k].  The
    caller starts each script's window (``begin_window``) itself: the loop is entered below ``DecodingTask.run``."""
    global _ROW_SCRIPTS
    _ROW_SCRIPTS = list(scripts) if scripts is not None else None


# ---------------------------------------------------------------------------
@torch.no_grad()
def detect_language(model, mel, tokenizer: Tokenizer = None):
    if tokenizer is None:
        tokenizer = get_tokenizer(model.is_multilingual, num_languages=model.num_languages)
    if tokenizer.language is None or tokenizer.language_token not in tokenizer.sot_sequence:
        raise ValueError("This model doesn't have language tokens so it can't perform lang id")
    single = mel.ndim == 2
    if single:
        mel = mel.unsqueeze(0)
    if mel.shape[-2:] != (model.dims.n_audio_ctx, model.dims.n_audio_state):
        mel = model.encoder(mel)
    n_audio = mel.shape[0]
    x = torch.tensor([[tokenizer.sot]] * n_audio).to(mel.device)
    logits = model.logits(x, mel)[:, 0]
    mask = torch.ones(logits.shape[-1], dtype=torch.bool)
    mask[list(tokenizer.all_language_tokens)] = False
    logits[:, mask] = -np.inf
    language_tokens = logits.argmax(dim=-1)
    probs = logits.softmax(dim=-1).cpu()
    language_probs = [{c: probs[i, j].item() for j, c in zip(tokenizer.all_language_tokens, tokenizer.all_language_codes)}
                      for i in range(n_audio)]
    if single:
        language_tokens, language_probs = language_tokens[0], language_probs[0]
    return language_tokens, language_probs


@dataclass(frozen=True)
class DecodingOptions:
    task: str = "transcribe"
    language: Optional[str] = None
    temperature: float = 0.0
    sample_len: Optional[int] = None
    best_of: Optional[int] = None
    beam_size: Optional[int] = None
    patience: Optional[float] = None
    length_penalty: Optional[float] = None
    prompt: Optional[Union[str, List[int]]] = None
    prefix: Optional[Union[str, List[int]]] = None
    suppress_tokens: Optional[Union[str, Iterable[int]]] = "-1"
    suppress_blank: bool = True
    without_timestamps: bool = False
    max_initial_timestamp: Optional[float] = 1.0
    fp16: bool = True


@dataclass(frozen=True)
class DecodingResult:
    audio_features: torch.Tensor
    language: str
    language_probs: Optional[dict] = None
    tokens: List[int] = field(default_factory=list)
    text: str = ""
    avg_logprob: float = np.nan
    no_speech_prob: float = np.nan
    temperature: float = np.nan
    compression_ratio: float = np.nan


class PyTorchInference:
    def __init__(self, model, initial_token_length: int):
        self.model = model
        self.initial_token_length = initial_token_length
        self.kv_cache = {}
        self.hooks = []
        self.kv_modules = [m for b in model.decoder.blocks for m in (b.attn.key, b.attn.value)]

    def logits(self, tokens, audio_features):
        if not self.kv_cache:
            self.kv_cache, self.hooks = self.model.install_kv_cache_hooks()
        if tokens.shape[-1] > self.initial_token_length:
            tokens = tokens[:, -1:]            # the cache holds everything before the last token
        return self.model.decoder(tokens, audio_features, kv_cache=self.kv_cache)

    def cleanup_caching(self):
        for h in self.hooks:
            h.remove()
        self.kv_cache, self.hooks = {}, []

    def rearrange_kv_cache(self, source_indices):
        if source_indices != list(range(len(source_indices))):
            for module in self.kv_modules:
                self.kv_cache[module] = self.kv_cache[module][source_indices].detach()


class MaximumLikelihoodRanker:
    def __init__(self, length_penalty: Optional[float]):
        self.length_penalty = length_penalty

    def rank(self, tokens, sum_logprobs):
        def scores(logprobs, lengths):
            out = []
            for lp, n in zip(logprobs, lengths):
                penalty = n if self.length_penalty is None else ((5 + n) / 6) ** self.length_penalty
                out.append(lp / penalty)
            return out
        lengths = [[len(t) for t in s] for s in tokens]
        return [int(np.argmax(scores(p, l))) for p, l in zip(sum_logprobs, lengths)]


class GreedyDecoder:
    def __init__(self, temperature: float, eot: int):
        self.temperature = temperature
        self.eot = eot

    def reset(self):
        pass

    def update(self, tokens, logits, sum_logprobs):
        if self.temperature == 0:
            next_tokens = logits.argmax(dim=-1)
        else:
            next_tokens = Categorical(logits=logits / self.temperature).sample()
        if _ROW_SCRIPTS is not None:
            assert len(_ROW_SCRIPTS) == tokens.shape[0], (len(_ROW_SCRIPTS), tokens.shape)
            done = (tokens[:, -1] == self.eot).tolist()     # a finished row is fed eot from now on: its script is over
            best = logits[:, :self.eot].argmax(dim=-1).tolist()     # every row's most likely text token: one host read
            next_tokens = torch.tensor([int(t) if (sc is None or done[k]) else sc.pick(int(t), None, self.eot, text_best=best[k])
                                        for k, (t, sc) in enumerate(zip(next_tokens.tolist(), _ROW_SCRIPTS))],
                                       device=logits.device)
        elif _SCRIPT is not None and tokens.shape[0] == 1:    # (several hypotheses: forced at the result level, see run())
            next_tokens = torch.tensor([_SCRIPT.pick(int(t), logits[k], self.eot) for k, t in enumerate(next_tokens.tolist())],
                                       device=logits.device)
        logprobs = F.log_softmax(logits.float(), dim=-1)
        current = logprobs[torch.arange(logprobs.shape[0]), next_tokens]
        sum_logprobs += current * (tokens[:, -1] != self.eot)
        next_tokens[tokens[:, -1] == self.eot] = self.eot
        tokens = torch.cat([tokens, next_tokens[:, None]], dim=-1)
        return tokens, bool((tokens[:, -1] == self.eot).all())

    def finalize(self, tokens, sum_logprobs):
        return F.pad(tokens, (0, 1), value=self.eot), sum_logprobs.tolist()


class BeamSearchDecoder:
    def __init__(self, beam_size: int, eot: int, inference: PyTorchInference, patience: Optional[float] = None):
        self.beam_size = beam_size
        self.eot = eot
        self.inference = inference
        self.patience = patience or 1.0
        self.max_candidates = round(beam_size * self.patience)
        self.finished_sequences = None
        assert self.max_candidates > 0, f"Invalid beam size ({beam_size}) or patience ({patience})"

    def reset(self):
        self.finished_sequences = None

    def update(self, tokens, logits, sum_logprobs):
        if tokens.shape[0] % self.beam_size != 0:
            raise ValueError(f"{tokens.shape}[0] % {self.beam_size} != 0")
        n_audio = tokens.shape[0] // self.beam_size
        if self.finished_sequences is None:
            self.finished_sequences = [{} for _ in range(n_audio)]
        logprobs = F.log_softmax(logits.float(), dim=-1)
        next_tokens, source_indices, finished = [], [], []
        for i in range(n_audio):
            scores, sources, done = {}, {}, {}
            for j in range(self.beam_size):           # expand every beam by its best beam_size+1 continuations
                idx = i * self.beam_size + j
                prefix = tokens[idx].tolist()
                for logprob, token in zip(*logprobs[idx].topk(self.beam_size + 1)):
                    seq = tuple(prefix + [token.item()])
                    scores[seq] = (sum_logprobs[idx] + logprob).item()
                    sources[seq] = idx
            saved = 0
            for seq in sorted(scores, key=scores.get, reverse=True):
                if seq[-1] == self.eot:
                    done[seq] = scores[seq]
                else:
                    sum_logprobs[len(next_tokens)] = scores[seq]
                    next_tokens.append(seq)
                    source_indices.append(sources[seq])
                    saved += 1
                    if saved == self.beam_size:
                        break
            finished.append(done)
        tokens = torch.tensor(next_tokens, device=tokens.device)
        self.inference.rearrange_kv_cache(source_indices)
        assert len(self.finished_sequences) == len(finished)
        for previously, newly in zip(self.finished_sequences, finished):
            for seq in sorted(newly, key=newly.get, reverse=True):
                if len(previously) >= self.max_candidates:
                    break
                previously[seq] = newly[seq]
        completed = all(len(s) >= self.max_candidates for s in self.finished_sequences)
        return tokens, completed

    def finalize(self, preceding_tokens, sum_logprobs):
        sum_logprobs = sum_logprobs.cpu()
        for i, sequences in enumerate(self.finished_sequences):
            if len(sequences) < self.beam_size:       # not enough finished beams: take the best unfinished ones
                for j in list(np.argsort(sum_logprobs[i]))[::-1]:
                    seq = preceding_tokens[i, j].tolist() + [self.eot]
                    sequences[tuple(seq)] = sum_logprobs[i][j].item()
                    if len(sequences) >= self.beam_size:
                        break
        tokens = [[torch.tensor(seq) for seq in sequences.keys()] for sequences in self.finished_sequences]
        sum_logprobs = [list(sequences.values()) for sequences in self.finished_sequences]
        return tokens, sum_logprobs


# ---------------------------------------------------------------------------
class SuppressBlank:
    def __init__(self, tokenizer: Tokenizer, sample_begin: int):
        self.tokenizer = tokenizer
        self.sample_begin = sample_begin

    def apply(self, logits, tokens):
        if tokens.shape[1] == self.sample_begin:
            logits[:, self.tokenizer.encode(" ") + [self.tokenizer.eot]] = -np.inf


class SuppressTokens:
    def __init__(self, suppress_tokens: Sequence[int]):
        self.suppress_tokens = list(suppress_tokens)

    def apply(self, logits, tokens):
        logits[:, self.suppress_tokens] = -np.inf


class ApplyTimestampRules:
    def __init__(self, tokenizer: Tokenizer, sample_begin: int, max_initial_timestamp_index: Optional[int]):
        self.tokenizer = tokenizer
        self.sample_begin = sample_begin
        self.max_initial_timestamp_index = max_initial_timestamp_index

    def apply(self, logits, tokens):
        tk = self.tokenizer
        ts0 = tk.timestamp_begin
        if tk.no_timestamps is not None:
            logits[:, tk.no_timestamps] = -np.inf
        for k in range(tokens.shape[0]):               # timestamps come in pairs, except right before eot
            sampled = tokens[k, self.sample_begin:]
            seq = sampled.tolist()
            last_was_ts = len(seq) >= 1 and seq[-1] >= ts0
            penultimate_was_ts = len(seq) < 2 or seq[-2] >= ts0
            if last_was_ts:
                if penultimate_was_ts:
                    logits[k, ts0:] = -np.inf          # a pair was just closed: text must follow
                else:
                    logits[k, : tk.eot] = -np.inf      # a timestamp must be followed by its twin or eot
            timestamps = sampled[sampled.ge(ts0)]
            if timestamps.numel() > 0:                 # timestamps never decrease
                if last_was_ts and not penultimate_was_ts:
                    timestamp_last = timestamps[-1]
                else:
                    timestamp_last = timestamps[-1] + 1
                logits[k, ts0:timestamp_last] = -np.inf
        if tokens.shape[1] == self.sample_begin:
            logits[:, :ts0] = -np.inf                  # the first sampled token is a timestamp
            if self.max_initial_timestamp_index is not None:
                logits[:, ts0 + self.max_initial_timestamp_index + 1:] = -np.inf
        logprobs = F.log_softmax(logits.float(), dim=-1)
        for k in range(tokens.shape[0]):               # total timestamp mass above any text token -> timestamp
            if logprobs[k, ts0:].logsumexp(dim=-1) > logprobs[k, :ts0].max():
                logits[k, :ts0] = -np.inf


# ---------------------------------------------------------------------------
class DecodingTask:
    def __init__(self, model, options: DecodingOptions):
        self.model = model
        language = options.language or "en"
        tokenizer = get_tokenizer(model.is_multilingual, num_languages=model.num_languages, language=language,
                                  task=options.task)
        self.tokenizer = tokenizer
        self.options = self._verify_options(options)
        self.n_group = options.beam_size or options.best_of or 1
        self.n_ctx = model.dims.n_text_ctx
        self.sample_len = options.sample_len or model.dims.n_text_ctx // 2
        self.sot_sequence = tokenizer.sot_sequence
        if self.options.without_timestamps:
            self.sot_sequence = tokenizer.sot_sequence_including_notimestamps
        self.initial_tokens = self._get_initial_tokens()
        self.sample_begin = len(self.initial_tokens)
        self.sot_index = self.initial_tokens.index(tokenizer.sot)
        self.inference = PyTorchInference(model, len(self.initial_tokens))
        self.sequence_ranker = MaximumLikelihoodRanker(options.length_penalty)
        if options.beam_size is not None:
            self.decoder = BeamSearchDecoder(options.beam_size, tokenizer.eot, self.inference, options.patience)
        else:
            self.decoder = GreedyDecoder(options.temperature, tokenizer.eot)
        self.logit_filters = []
        if self.options.suppress_blank:
            self.logit_filters.append(SuppressBlank(self.tokenizer, self.sample_begin))
        if self.options.suppress_tokens:
            self.logit_filters.append(SuppressTokens(self._get_suppress_tokens()))
        if not options.without_timestamps:
            precision = CHUNK_LENGTH / model.dims.n_audio_ctx
            max_initial = None
            if options.max_initial_timestamp:
                max_initial = round(self.options.max_initial_timestamp / precision)
            self.logit_filters.append(ApplyTimestampRules(tokenizer, self.sample_begin, max_initial))

    def _verify_options(self, options):
        if options.beam_size is not None and options.best_of is not None:
            raise ValueError("beam_size and best_of can't be given together")
        if options.temperature == 0 and options.best_of is not None:
            raise ValueError("best_of with greedy sampling (T=0) is not compatible")
        if options.patience is not None and options.beam_size is None:
            raise ValueError("patience requires beam_size to be given")
        if options.length_penalty is not None and not (0 <= options.length_penalty <= 1):
            raise ValueError("length_penalty (alpha) should be a value between 0 and 1")
        return options

    def _get_initial_tokens(self) -> Tuple[int]:
        tokens = list(self.sot_sequence)
        if prefix := self.options.prefix:
            prefix_tokens = self.tokenizer.encode(" " + prefix.strip()) if isinstance(prefix, str) else prefix
            if self.sample_len is not None:
                prefix_tokens = prefix_tokens[-(self.n_ctx // 2 - self.sample_len):]
            tokens = tokens + prefix_tokens
        if prompt := self.options.prompt:
            prompt_tokens = self.tokenizer.encode(" " + prompt.strip()) if isinstance(prompt, str) else prompt
            tokens = [self.tokenizer.sot_prev] + prompt_tokens[-(self.n_ctx // 2 - 1):] + tokens
        return tuple(tokens)

    def _get_suppress_tokens(self) -> Tuple[int]:
        suppress = self.options.suppress_tokens
        if isinstance(suppress, str):
            suppress = [int(t) for t in suppress.split(",")]
        if -1 in suppress:
            suppress = [t for t in suppress if t >= 0]
            suppress.extend(self.tokenizer.non_speech_tokens)
        elif suppress is None or len(suppress) == 0:
            suppress = []
        else:
            assert isinstance(suppress, list), "suppress_tokens must be a list"
        tk = self.tokenizer
        suppress.extend([tk.transcribe, tk.translate, tk.sot, tk.sot_prev, tk.sot_lm])
        if tk.no_speech is not None:
            suppress.append(tk.no_speech)
        return tuple(sorted(set(suppress)))

    def _get_audio_features(self, mel):
        if self.options.fp16:
            mel = mel.half()
        if mel.shape[-2:] == (self.model.dims.n_audio_ctx, self.model.dims.n_audio_state):
            audio_features = mel
        else:
            audio_features = self.model.encoder(mel)
        if audio_features.dtype != (torch.float16 if self.options.fp16 else torch.float32):
            raise TypeError(f"audio_features has an incorrect dtype: {audio_features.dtype}")
        return audio_features

    def _detect_language(self, audio_features, tokens):
        languages = [self.options.language] * audio_features.shape[0]
        lang_probs = None
        if self.options.language is None or self.options.task == "lang_id":
            lang_tokens, lang_probs = self.model.detect_language(audio_features, self.tokenizer)
            languages = [max(probs, key=probs.get) for probs in lang_probs]
            if self.options.language is None:
                tokens[:, self.sot_index + 1] = lang_tokens
        return languages, lang_probs

    def _main_loop(self, audio_features, tokens):
        n_batch = tokens.shape[0]
        sum_logprobs = torch.zeros(n_batch, device=audio_features.device)
        no_speech_probs = [np.nan] * n_batch
        try:
            for i in range(self.sample_len):
                logits = self.inference.logits(tokens, audio_features)
                if i == 0 and self.tokenizer.no_speech is not None:
                    probs_at_sot = logits[:, self.sot_index].float().softmax(dim=-1)
                    no_speech_probs = probs_at_sot[:, self.tokenizer.no_speech].tolist()
                logits = logits[:, -1]
                for logit_filter in self.logit_filters:
                    logit_filter.apply(logits, tokens)
                tokens, completed = self.decoder.update(tokens, logits, sum_logprobs)
                if completed or tokens.shape[-1] > self.n_ctx:
                    break
        finally:
            self.inference.cleanup_caching()
        return tokens, sum_logprobs, no_speech_probs

    @torch.no_grad()
    def run(self, mel) -> List[DecodingResult]:
        self.decoder.reset()
        if _SCRIPT is not None:
            _SCRIPT.begin_window()
        tokenizer = self.tokenizer
        n_audio = mel.shape[0]
        audio_features = self._get_audio_features(mel)
        tokens = torch.tensor([self.initial_tokens]).repeat(n_audio, 1)
        languages, language_probs = self._detect_language(audio_features, tokens)
        if self.options.task == "lang_id":
            return [DecodingResult(audio_features=f, language=l, language_probs=p)
                    for f, l, p in zip(audio_features, languages, language_probs)]
        tokens = tokens.repeat_interleave(self.n_group, dim=0).to(audio_features.device)
        tokens, sum_logprobs, no_speech_probs = self._main_loop(audio_features, tokens)
        audio_features = audio_features[:: self.n_group]
        no_speech_probs = no_speech_probs[:: self.n_group]
        assert audio_features.shape[0] == len(no_speech_probs) == n_audio
        tokens = tokens.reshape(n_audio, self.n_group, -1)
        sum_logprobs = sum_logprobs.reshape(n_audio, self.n_group)
        tokens, sum_logprobs = self.decoder.finalize(tokens, sum_logprobs)
        tokens = [[t[self.sample_begin: (t == tokenizer.eot).nonzero()[0, 0]] for t in s] for s in tokens]
        selected = self.sequence_ranker.rank(tokens, sum_logprobs)
        tokens = [t[i].tolist() for i, t in zip(selected, tokens)]
        if _SCRIPT is not None and self.n_group > 1:      # beam search / best_of: the scripted window replaces the result
            tokens = [_SCRIPT.final_tokens(t, tokenizer.eot) for t in tokens]
        texts = [tokenizer.decode(t).strip() for t in tokens]
        sum_logprobs = [lp[i] for i, lp in zip(selected, sum_logprobs)]
        avg_logprobs = [lp / (len(t) + 1) for t, lp in zip(tokens, sum_logprobs)]
        fields = (texts, languages, tokens, audio_features, avg_logprobs, no_speech_probs)
        if len(set(map(len, fields))) != 1:
            raise RuntimeError(f"inconsistent result lengths: {list(map(len, fields))}")
        return [DecodingResult(audio_features=features, language=language, tokens=tokens, text=text, avg_logprob=avg_logprob,
                               no_speech_prob=no_speech_prob, temperature=self.options.temperature,
                               compression_ratio=compression_ratio(text))
                for text, language, tokens, features, avg_logprob, no_speech_prob in zip(*fields)]


@torch.no_grad()
def decode(model, mel, options: DecodingOptions = DecodingOptions(), **kwargs):
    if single := mel.ndim == 2:
        mel = mel.unsqueeze(0)
    if kwargs:
        options = replace(options, **kwargs)
    result = DecodingTask(model, options).run(mel)
    return result[0] if single else result
