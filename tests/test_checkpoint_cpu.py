"""HF-format checkpoint conversion (whisper_timestamped/checkpoint.py), and -- as a by-product -- an independent
pin of the whisper test double: a random `transformers.WhisperForConditionalGeneration`, converted through the
reference's key renaming, must produce the same logits and cross-attention as transformers' own forward."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hf_model():
    import transformers
    cfg = transformers.WhisperConfig(vocab_size=51865, num_mel_bins=80, d_model=128, encoder_layers=2, decoder_layers=3,
                                     encoder_attention_heads=2, decoder_attention_heads=2, encoder_ffn_dim=512,
                                     decoder_ffn_dim=512, max_source_positions=1500, max_target_positions=448,
                                     attn_implementation="eager")
    torch.manual_seed(0)
    return transformers.WhisperForConditionalGeneration(cfg).eval()


@pytest.mark.parametrize("tied", [False, True])
def test_convert_hf_state_dict_matches_transformers_forward(hf_model, tied):
    import whisper_double as W
    W.install()
    from whisper_timestamped.checkpoint import convert_hf_state_dict, hf_to_whisper_states, states_to_dim
    sd = {k: v.clone() for k, v in hf_model.state_dict().items()}
    if tied:
        sd.pop("proj_out.weight")
    model = convert_hf_state_dict(sd, device="cpu").eval()
    assert type(model).__name__ == ("Whisper" if tied else "WhisperUntied")
    d = model.dims
    assert (d.n_text_layer, d.n_audio_layer, d.n_text_state, d.n_text_head, d.n_vocab, d.n_mels) == (3, 2, 128, 2, 51865, 80)
    assert not hasattr(model, "alignment_heads")
    assert hf_to_whisper_states("model.decoder.layers.2.encoder_attn.q_proj.weight") == "decoder.blocks.2.cross_attn.query.weight"
    assert hf_to_whisper_states("base_model.model.model.encoder.layer_norm.bias") == "encoder.ln_post.bias"
    assert hf_to_whisper_states("_mel_filters") is None

    g = torch.Generator().manual_seed(1)
    mel = torch.randn(1, 80, 3000, generator=g)
    tokens = torch.tensor([[50258, 50259, 50359, 50364, 6455, 11, 2232, 50464]])
    with torch.no_grad():
        ref = hf_model(input_features=mel, decoder_input_ids=tokens, output_attentions=True)
        with W.model.disable_sdpa():
            qks = []
            hooks = [b.cross_attn.register_forward_hook(lambda m, i, o: qks.append(o[1])) for b in model.decoder.blocks]
            got = model(mel, tokens)
            for h in hooks:
                h.remove()
    np.testing.assert_allclose(got.numpy(), ref.logits.numpy(), rtol=2e-4, atol=2e-4)
    # cross-attention: softmax(QK logits of the double) == transformers' attention probabilities
    for qk, att in zip(qks, ref.cross_attentions):
        np.testing.assert_allclose(qk.softmax(-1).numpy(), att.numpy(), rtol=1e-4, atol=1e-6)
