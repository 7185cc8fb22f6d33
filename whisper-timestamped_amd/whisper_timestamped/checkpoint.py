"""Loading HuggingFace-format Whisper checkpoints into an openai-whisper model.

/root/reference/whisper_timestamped/transcribe.py:2478-2544 (file discovery + conversion inside load_model),
:2546-2564 (torch_load), :2876-2906 (key renaming), :2909-2923 (dimensions from the tensors),
:2925-2962 (untied output projection).  Host-side plumbing only; nothing here touches the GPU kernels.
"""
from __future__ import annotations

import json
import logging
import os
import re

import torch

from . import backend as _backend

logger = logging.getLogger("whisper_timestamped")

# HuggingFace name fragment -> openai-whisper name fragment, applied in this order.  "[._]": the reference's patterns
# use an unescaped '.', which is what turns "self_attn_layer_norm" into "attn.layer_norm" (-> "attn_ln") and
# "encoder_attn_layer_norm" into "cross_attn.layer_norm" (-> "cross_attn_ln").
_RENAMES = [
    (r"\.layers\.", ".blocks."), (r"\.self_attn[._]", ".attn."), (r"\.q_proj\.", ".query."), (r"\.k_proj\.", ".key."),
    (r"\.v_proj\.", ".value."), (r"\.out_proj\.", ".out."), (r"\.fc1\.", ".mlp.0."), (r"\.fc2\.", ".mlp.2."),
    (r"\.fc3\.", ".mlp.3."), (r"\.encoder_attn[._]", ".cross_attn."), (r"\.cross_attn\.ln\.", ".cross_attn_ln."),
    (r"\.embed_positions\.weight", ".positional_embedding"), (r"\.embed_tokens\.", ".token_embedding."),
    (r"model\.", ""), (r"attn\.layer_norm\.", "attn_ln."), (r"\.final_layer_norm\.", ".mlp_ln."),
    (r"encoder\.layer_norm\.", "encoder.ln_post."), (r"decoder\.layer_norm\.", "decoder.ln."),
]


def hf_to_whisper_states(name: str):
    """HF parameter name -> openai-whisper parameter name (None = drop)."""
    if name == "_mel_filters":                 # speechbrain
        return None
    if "default" in name:                      # PEFT adapters
        return None
    if name.startswith("base_model.model."):
        name = name[len("base_model.model."):]
    for pattern, repl in _RENAMES:
        name = re.sub(pattern, repl, name)
    return name


def _count_blocks(state_dict, prefix):
    return len({".".join(k.split(".")[:3]) for k in state_dict if prefix in k})


def states_to_dim(state_dict):
    n_audio_state = len(state_dict["encoder.ln_post.bias"])
    n_text_state = len(state_dict["decoder.ln.bias"])
    return dict(
        n_mels=state_dict["encoder.conv1.weight"].shape[1],
        n_vocab=state_dict["decoder.token_embedding.weight"].shape[0],
        n_audio_ctx=state_dict["encoder.positional_embedding"].shape[0],
        n_audio_state=n_audio_state, n_audio_head=n_audio_state // 64,
        n_audio_layer=_count_blocks(state_dict, "encoder.blocks."),
        n_text_ctx=state_dict["decoder.positional_embedding"].shape[0],
        n_text_state=n_text_state, n_text_head=n_text_state // 64,
        n_text_layer=_count_blocks(state_dict, "decoder.blocks."),
    )


def torch_load(model_path):
    if isinstance(model_path, list):
        merged = {}
        for p in model_path:
            part = torch_load(p)
            for k in part:
                assert k not in merged, f"Found duplicate key {k} in {p}"
            merged.update(part)
        return merged
    assert isinstance(model_path, str)
    if model_path.endswith(".safetensors"):
        from safetensors import safe_open
        out = {}
        with safe_open(model_path, framework="pt", device="cpu") as f:
            for k in f.keys():
                out[k] = f.get_tensor(k)
        return out
    return torch.load(model_path, map_location="cpu")


def find_checkpoint_files(name, download_root=None):
    """Local folder / file, else the HuggingFace cache (needs `transformers`; no download happens offline)."""
    ext = os.path.splitext(name)[-1] if os.path.isfile(name) else None
    if ext in (".ckpt", ".bin", ".safetensors"):
        return name
    if os.path.isdir(name):
        for candidate in ("pytorch_model.bin", "whisper.ckpt", "model.safetensors"):
            path = os.path.join(name, candidate)
            if os.path.isfile(path):
                return path
        for index in ("pytorch_model.bin.index.json", "model.safetensors.index.json"):
            path = os.path.join(name, index)
            if os.path.isfile(path):
                mapping = json.load(open(path))
                assert isinstance(mapping.get("weight_map"), dict)
                return [os.path.join(name, p) for p in sorted(set(mapping["weight_map"].values()))]
    try:
        from transformers.utils import cached_file
    except ImportError:
        raise ImportError(f"If you are trying to download a HuggingFace model with {name}, please install first the transformers library")
    kwargs = dict(cache_dir=os.path.join(download_root, "huggingface", "hub") if download_root else None, revision=None)
    last = None
    for candidate in ("pytorch_model.bin", "whisper.ckpt", "pytorch_model.bin.index.json", "model.safetensors",
                      "model.safetensors.index.json"):
        try:
            path = cached_file(name, candidate, **kwargs)
        except OSError as err:
            last = err
            continue
        if candidate.endswith("index.json"):
            mapping = json.load(open(path))
            assert isinstance(mapping.get("weight_map"), dict)
            folder = os.path.dirname(path)
            return [os.path.join(folder, p) for p in sorted(set(mapping["weight_map"].values()))]
        return path
    raise RuntimeError(f"Original error: {last}\nCould not find model {name} from HuggingFace nor local folders.")


def _untied_class():
    """A Whisper whose output projection is not tied to the token embedding (fine-tuned HF checkpoints)."""
    w = _backend.whisper()

    class TextDecoderUntied(w.model.TextDecoder):
        def __init__(self, *args, **kwargs):
            super().__init__(*args, **kwargs)
            n_vocab, n_state = self.token_embedding.weight.shape
            self.proj_out = torch.nn.Linear(n_state, n_vocab, bias=False)

        def forward(self, x, xa, kv_cache=None):
            offset = next(iter(kv_cache.values())).shape[1] if kv_cache else 0
            x = self.token_embedding(x) + self.positional_embedding[offset: offset + x.shape[-1]]
            x = x.to(xa.dtype)
            for block in self.blocks:
                x = block(x, xa, mask=self.mask, kv_cache=kv_cache)
            x = self.ln(x)
            return self.proj_out.to(x.dtype)(x).float()

    class WhisperUntied(w.model.Whisper):
        def __init__(self, *args, **kwargs):
            super().__init__(*args, **kwargs)
            d = self.dims
            self.decoder = TextDecoderUntied(d.n_vocab, d.n_text_ctx, d.n_text_state, d.n_text_head, d.n_text_layer)

    return WhisperUntied


def convert_hf_state_dict(hf_state_dict, device=None):
    """HF state dict (tensors on the CPU) -> openai-whisper model on `device`."""
    w = _backend.whisper()
    sd = {}
    for key, tensor in hf_state_dict.items():
        new_key = hf_to_whisper_states(key)
        if new_key is not None:
            sd[new_key] = tensor
    dims = w.model.ModelDimensions(**states_to_dim(sd))
    if "proj_out.weight" in sd:
        sd["decoder.proj_out.weight"] = sd.pop("proj_out.weight")
        logger.warning("Using untied projection layer")
        model = _untied_class()(dims)
    else:
        model = w.model.Whisper(dims)
    model.load_state_dict(sd)
    if hasattr(model, "alignment_heads"):
        del model.alignment_heads          # recomputed by get_alignment_heads (parameter-count table)
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    return model.to(device)
