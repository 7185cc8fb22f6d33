"""SURVEY.md 8(f) N4 on the GPU: checkpoints from FILES (openai-whisper ``.pt``: {"dims", "model_state_dict"}; HuggingFace
``safetensors`` with the key renaming and the untied output projection of
/root/reference/whisper_timestamped/transcribe.py:2405-2564, 2876-2962), the repository's own tokenizer through
$WT_TOKENIZER_VOCAB (whisper_timestamped/vocab.py), and the command line (transcribe.py:2964-3182) -- each against the same
model built in memory: identical words, times and confidences (the kernels are the real ones; weights are random, the
sampler scripted)."""
import dataclasses
import json
import os

import numpy as np
import pytest
import torch

import schema_check
from golden import make_golden_transcribe as G

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ML, EOT = 50364, 50257


def _audio(seconds, seed):
    g = torch.Generator().manual_seed(seed)
    n = int(seconds * 16000)
    t = torch.arange(n) / 16000.0
    return (0.05 * torch.randn(n, generator=g) + 0.1 * torch.sin(2 * np.pi * 220.0 * t)).float()


def _script(ids=None):
    """One window of four segments on the peaked double's ridge; `ids`: the text tokens to use (default: G.text_ids)."""
    import many_helper as H
    counts = [9, 14, 7, 12]
    if ids is not None:
        assert len(ids) >= sum(counts)
        it = iter(ids)
    segs = [(s, G.text_ids(900 + k, n) if ids is None else [next(it) for _ in range(n)], e)
            for k, (s, n, e) in enumerate(H.peaked_segments(counts))]
    return [G.window_script(ML, EOT, segs, "eot")]


def _transcribe(model, audio, ids=None, **kw):
    from whisper_double.decoding import Script, set_script
    import whisper_timestamped as wt
    set_script(Script(_script(ids)))
    try:
        return wt.transcribe(model, audio, language="en", fp16=False, **kw)
    finally:
        set_script(None)


def _words(r):
    return [(w["text"], w["start"], w["end"], w["confidence"]) for s in r["segments"] for w in s["words"]]


@pytest.fixture(scope="module")
def vocab_dir(tmp_path_factory):
    from test_tokenizer_cpu import train_ranks, write_vocab
    d = tmp_path_factory.mktemp("vocab")
    write_vocab(d / "multilingual.tiktoken", train_ranks(400, pad_to=50257))
    return str(d)


def test_pt_checkpoint_file_through_load_model_and_the_own_tokenizer(tmp_path, vocab_dir, monkeypatch):
    import whisper_double as W
    W.install()
    import whisper_timestamped as wt
    from whisper_timestamped import vocab
    memory = W.build_model("tiny", seed=5, device="cuda:0", attention="peaked")
    path = str(tmp_path / "tiny-random.pt")
    torch.save({"dims": dataclasses.asdict(memory.dims), "model_state_dict": {k: v.cpu() for k, v in memory.state_dict().items()}}, path)
    loaded = wt.load_model(path, device="cuda:0")
    assert loaded.device.type == "cuda" and loaded.dims == memory.dims
    audio = _audio(13.0, 3)
    # the backend's tokenizer (the double's) and the repository's own, loaded from a .tiktoken file
    ref = _transcribe(memory, audio)
    got = _transcribe(loaded, audio)
    assert len(_words(ref)) > 20 and _words(got) == _words(ref) and got["text"] == ref["text"]
    monkeypatch.setenv("WT_TOKENIZER_VOCAB", vocab_dir)
    from whisper_timestamped import backend
    own_tk = backend.get_tokenizer(loaded, task="transcribe", language="en")
    assert isinstance(own_tk, vocab.Tokenizer)
    # a transcript IN that vocabulary (ids of real text; most ids of the 400-merge test vocabulary are filler bytes that
    # never complete a character, which the reference's word splitter swallows whole)
    text = " Let's go with it again and again, you know. So we went there, and it was fine! Then what? Nothing: we came back."
    banned = set(W.tokenizer.get_tokenizer(True, language="en").non_speech_tokens) | {220}     # what the backend's sampler suppresses
    ids = [t for t in own_tk.encode(text) if t not in banned]
    own_ref = _transcribe(memory, audio, ids=ids)
    own = _transcribe(loaded, audio, ids=ids)
    assert len(_words(own)) > 8 and _words(own) == _words(own_ref)
    said = " ".join(w[0] for w in _words(own))
    assert "Let's go with it again" in said, said                   # words of the own vocabulary, split by the reference's rules
    assert all(t >= own_tk.timestamp_begin for s_ in own["segments"] for t in (s_["tokens"][0], s_["tokens"][-1]))
    schema_check.validate(own, json.load(open(os.path.join(HERE, "golden", "json_schema.json"))))


@pytest.mark.parametrize("tied", [False, True])
def test_hf_safetensors_checkpoint_through_load_model(tmp_path, tied):
    transformers = pytest.importorskip("transformers")
    import whisper_double as W
    W.install()
    import whisper_timestamped as wt
    from whisper_timestamped.checkpoint import convert_hf_state_dict
    cfg = transformers.WhisperConfig(vocab_size=51865, num_mel_bins=80, d_model=384, encoder_layers=4, decoder_layers=4,
                                     encoder_attention_heads=6, decoder_attention_heads=6, encoder_ffn_dim=1536,
                                     decoder_ffn_dim=1536, max_source_positions=1500, max_target_positions=448,
                                     tie_word_embeddings=tied)
    torch.manual_seed(3)
    hf = transformers.WhisperForConditionalGeneration(cfg).eval()
    with torch.no_grad():
        # scripted decoding needs what whisper_double.build_model arranges: text logits spread wide, timestamp logits narrow
        # (else the sampler's "timestamps dominate" rule suppresses every text token and their log-probabilities are -inf)
        head = torch.randn(hf.proj_out.weight.shape) * 0.25
        head[ML:] *= 0.16
        hf.model.decoder.embed_tokens.weight.copy_(head)
        if not tied:                                                 # a head of its own (fine-tuned checkpoints)
            other = torch.randn(hf.proj_out.weight.shape) * 0.25
            other[ML:] *= 0.16
            hf.proj_out.weight.copy_(other)
            assert not torch.equal(hf.proj_out.weight, hf.model.decoder.embed_tokens.weight)
    folder = tmp_path / ("hf_tied" if tied else "hf_untied")
    hf.save_pretrained(str(folder), safe_serialization=True)
    assert (folder / "model.safetensors").is_file()
    loaded = wt.load_model(str(folder), device="cuda:0")
    assert type(loaded).__name__ == ("Whisper" if tied else "WhisperUntied") and loaded.device.type == "cuda"
    sd = {k: v.clone() for k, v in hf.state_dict().items()}
    if tied:
        sd.pop("proj_out.weight", None)
    memory = convert_hf_state_dict(sd, device="cuda:0")
    audio = _audio(11.0, 4)
    ref, got = _transcribe(memory, audio), _transcribe(loaded, audio)
    assert len(_words(ref)) > 20 and _words(got) == _words(ref)
    # the heads come from the parameter-count table or the top-layers fallback: the attribute is gone after conversion
    assert not hasattr(loaded, "alignment_heads")
    # the file on its own (not a folder), and the naive strategy over the loaded model
    again = wt.load_model(str(folder / "model.safetensors"), device="cuda:0")
    assert _words(_transcribe(again, audio)) == _words(ref)
    naive_ref, naive = _transcribe(memory, audio, naive_approach=True), _transcribe(loaded, audio, naive_approach=True)
    assert len(_words(naive)) > 20 and _words(naive) == _words(naive_ref)


def test_cli_on_the_gpu_writes_every_format_from_a_checkpoint_file(tmp_path, monkeypatch):
    from scipy.io import wavfile
    import whisper_double as W
    from whisper_double.decoding import Script, set_script
    W.install()
    from whisper_timestamped import cli as C
    memory = W.build_model("tiny", seed=6, device="cuda:0", attention="peaked")
    path = str(tmp_path / "tiny-random.pt")
    torch.save({"dims": dataclasses.asdict(memory.dims), "model_state_dict": {k: v.cpu() for k, v in memory.state_dict().items()}}, path)
    audio = _audio(12.0, 8)
    wav = tmp_path / "clip.wav"
    wavfile.write(str(wav), 16000, (audio.numpy() * 32767).astype(np.int16))
    out = tmp_path / "out"
    set_script(Script(_script()))
    try:
        C.cli([str(wav), "--model", path, "--device", "cuda:0", "--language", "en", "--output_dir", str(out), "--fp16", "False"])
    finally:
        set_script(None)
    names = sorted(os.listdir(out))
    assert names == sorted("clip.wav" + s for s in (".words.json", ".txt", ".vtt", ".words.vtt", ".srt", ".words.srt", ".csv",
                                                    ".words.csv", ".tsv", ".words.tsv"))
    result = json.load(open(out / "clip.wav.words.json", encoding="utf-8"))
    schema_check.validate(result, json.load(open(os.path.join(HERE, "golden", "json_schema.json"))))
    # the Python API on the same file with the in-memory model: the same words (the CLI reads the .wav: int16 quantisation on both sides)
    from whisper_timestamped.audio import load_audio
    ref = _transcribe(memory, torch.from_numpy(load_audio(str(wav))))
    assert len(_words(ref)) > 20 and _words(result) == _words(ref)
    n_words = len(_words(result))
    assert (out / "clip.wav.words.srt").read_text().count(" --> ") == n_words
    assert (out / "clip.wav.vtt").read_text().startswith("WEBVTT\n")
