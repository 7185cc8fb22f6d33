"""whisper.model stand-in: the Whisper encoder/decoder.

Module names match openai-whisper's state-dict keys (``encoder.conv1``,
``decoder.token_embedding``, ``decoder.blocks[i].cross_attn.query`` ...), because
whisper-timestamped addresses the model through them.
"""
from contextlib import contextmanager
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .decoding import decode as decode_function
from .decoding import detect_language as detect_language_function
from .transcribe import transcribe as transcribe_function


@dataclass
class ModelDimensions:
    n_mels: int
    n_audio_ctx: int
    n_audio_state: int
    n_audio_head: int
    n_audio_layer: int
    n_vocab: int
    n_text_ctx: int
    n_text_state: int
    n_text_head: int
    n_text_layer: int


class LayerNorm(nn.LayerNorm):
    def forward(self, x):
        return super().forward(x.float()).type(x.dtype)


class Linear(nn.Linear):
    def forward(self, x):
        return F.linear(x, self.weight.to(x.dtype), None if self.bias is None else self.bias.to(x.dtype))


class Conv1d(nn.Conv1d):
    def _conv_forward(self, x, weight, bias):
        return super()._conv_forward(x, weight.to(x.dtype), None if bias is None else bias.to(x.dtype))


def sinusoids(length, channels, max_timescale=10000):
    assert channels % 2 == 0
    log_inc = np.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-log_inc * torch.arange(channels // 2))
    t = torch.arange(length)[:, None] * inv[None, :]
    return torch.cat([torch.sin(t), torch.cos(t)], dim=1)


@contextmanager
def disable_sdpa():
    prev = MultiHeadAttention.use_sdpa
    try:
        MultiHeadAttention.use_sdpa = False
        yield
    finally:
        MultiHeadAttention.use_sdpa = prev


class MultiHeadAttention(nn.Module):
    use_sdpa = True

    def __init__(self, n_state: int, n_head: int):
        super().__init__()
        self.n_head = n_head
        self.query = Linear(n_state, n_state)
        self.key = Linear(n_state, n_state, bias=False)
        self.value = Linear(n_state, n_state)
        self.out = Linear(n_state, n_state)

    def forward(self, x, xa=None, mask=None, kv_cache=None):
        q = self.query(x)
        if kv_cache is None or xa is None or self.key not in kv_cache:
            src = x if xa is None else xa
            k, v = self.key(src), self.value(src)
        else:                      # cross-attention keys/values of this window are already cached
            k, v = kv_cache[self.key], kv_cache[self.value]
        wv, qk = self.qkv_attention(q, k, v, mask)
        return self.out(wv), qk

    def qkv_attention(self, q, k, v, mask=None):
        n_batch, n_ctx, n_state = q.shape
        scale = (n_state // self.n_head) ** -0.25
        q = q.view(*q.shape[:2], self.n_head, -1).permute(0, 2, 1, 3)
        k = k.view(*k.shape[:2], self.n_head, -1).permute(0, 2, 1, 3)
        v = v.view(*v.shape[:2], self.n_head, -1).permute(0, 2, 1, 3)
        if MultiHeadAttention.use_sdpa:
            a = F.scaled_dot_product_attention(q, k, v, is_causal=mask is not None and n_ctx > 1)
            return a.permute(0, 2, 1, 3).flatten(start_dim=2), None
        qk = (q * scale) @ (k * scale).transpose(-1, -2)
        if mask is not None:
            qk = qk + mask[:n_ctx, :n_ctx]
        qk = qk.float()
        w = F.softmax(qk, dim=-1).to(q.dtype)
        return (w @ v).permute(0, 2, 1, 3).flatten(start_dim=2), qk.detach()


class ResidualAttentionBlock(nn.Module):
    def __init__(self, n_state: int, n_head: int, cross_attention: bool = False):
        super().__init__()
        self.attn = MultiHeadAttention(n_state, n_head)
        self.attn_ln = LayerNorm(n_state)
        self.cross_attn = MultiHeadAttention(n_state, n_head) if cross_attention else None
        self.cross_attn_ln = LayerNorm(n_state) if cross_attention else None
        self.mlp = nn.Sequential(Linear(n_state, n_state * 4), nn.GELU(), Linear(n_state * 4, n_state))
        self.mlp_ln = LayerNorm(n_state)

    def forward(self, x, xa=None, mask=None, kv_cache=None):
        x = x + self.attn(self.attn_ln(x), mask=mask, kv_cache=kv_cache)[0]
        if self.cross_attn:
            x = x + self.cross_attn(self.cross_attn_ln(x), xa, kv_cache=kv_cache)[0]
        return x + self.mlp(self.mlp_ln(x))


class AudioEncoder(nn.Module):
    def __init__(self, n_mels, n_ctx, n_state, n_head, n_layer):
        super().__init__()
        self.conv1 = Conv1d(n_mels, n_state, kernel_size=3, padding=1)
        self.conv2 = Conv1d(n_state, n_state, kernel_size=3, stride=2, padding=1)
        self.register_buffer("positional_embedding", sinusoids(n_ctx, n_state))
        self.blocks = nn.ModuleList([ResidualAttentionBlock(n_state, n_head) for _ in range(n_layer)])
        self.ln_post = LayerNorm(n_state)

    def forward(self, x):
        x = F.gelu(self.conv1(x))
        x = F.gelu(self.conv2(x))
        x = x.permute(0, 2, 1)
        assert x.shape[1:] == self.positional_embedding.shape, "incorrect audio shape"
        x = (x + self.positional_embedding).to(x.dtype)
        for block in self.blocks:
            x = block(x)
        return self.ln_post(x)


class TextDecoder(nn.Module):
    def __init__(self, n_vocab, n_ctx, n_state, n_head, n_layer):
        super().__init__()
        self.token_embedding = nn.Embedding(n_vocab, n_state)
        self.positional_embedding = nn.Parameter(torch.empty(n_ctx, n_state))
        self.blocks = nn.ModuleList([ResidualAttentionBlock(n_state, n_head, cross_attention=True) for _ in range(n_layer)])
        self.ln = LayerNorm(n_state)
        self.register_buffer("mask", torch.empty(n_ctx, n_ctx).fill_(-np.inf).triu_(1), persistent=False)

    def forward(self, x, xa, kv_cache=None):
        offset = next(iter(kv_cache.values())).shape[1] if kv_cache else 0
        x = self.token_embedding(x) + self.positional_embedding[offset: offset + x.shape[-1]]
        x = x.to(xa.dtype)
        for block in self.blocks:
            x = block(x, xa, mask=self.mask, kv_cache=kv_cache)
        x = self.ln(x)
        return (x @ torch.transpose(self.token_embedding.weight.to(x.dtype), 0, 1)).float()


class Whisper(nn.Module):
    def __init__(self, dims: ModelDimensions):
        super().__init__()
        self.dims = dims
        self.encoder = AudioEncoder(dims.n_mels, dims.n_audio_ctx, dims.n_audio_state, dims.n_audio_head, dims.n_audio_layer)
        self.decoder = TextDecoder(dims.n_vocab, dims.n_text_ctx, dims.n_text_state, dims.n_text_head, dims.n_text_layer)
        # default alignment heads: the upper half of the decoder layers, all heads
        heads = torch.zeros(dims.n_text_layer, dims.n_text_head, dtype=torch.bool)
        heads[dims.n_text_layer // 2:] = True
        self.register_buffer("alignment_heads", heads.to_sparse(), persistent=False)

    def set_alignment_heads(self, dump: bytes):
        import base64
        import gzip
        array = np.frombuffer(gzip.decompress(base64.b85decode(dump)), dtype=bool).copy()
        mask = torch.from_numpy(array).reshape(self.dims.n_text_layer, self.dims.n_text_head)
        self.register_buffer("alignment_heads", mask.to_sparse(), persistent=False)

    def embed_audio(self, mel):
        return self.encoder(mel)

    def logits(self, tokens, audio_features):
        return self.decoder(tokens, audio_features)

    def forward(self, mel, tokens):
        return self.decoder(tokens, self.encoder(mel))

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def is_multilingual(self):
        return self.dims.n_vocab >= 51865

    @property
    def num_languages(self):
        return self.dims.n_vocab - 51765 - int(self.is_multilingual)

    def install_kv_cache_hooks(self, cache=None):
        cache = dict(cache) if cache is not None else {}
        hooks = []

        def save_to_cache(module, _, output):
            if module not in cache or output.shape[1] > self.dims.n_text_ctx:
                cache[module] = output          # first token, or cross-attention (1500 > n_text_ctx): keep as is
            else:
                cache[module] = torch.cat([cache[module], output], dim=1).detach()
            return cache[module]

        def install(layer):
            if isinstance(layer, MultiHeadAttention):
                hooks.append(layer.key.register_forward_hook(save_to_cache))
                hooks.append(layer.value.register_forward_hook(save_to_cache))

        self.decoder.apply(install)
        return cache, hooks

    detect_language = detect_language_function
    transcribe = transcribe_function
    decode = decode_function


_DIMS = {  # name: (n_mels, audio_state, audio_head, audio_layer, text_state, text_head, text_layer)
    "tiny": (80, 384, 6, 4, 384, 6, 4), "base": (80, 512, 8, 6, 512, 8, 6), "small": (80, 768, 12, 12, 768, 12, 12),
    "medium": (80, 1024, 16, 24, 1024, 16, 24), "large-v3": (128, 1280, 20, 32, 1280, 20, 32),
    # tiny-sized model with large-v3's front end and vocabulary (128 mels, 100 languages, specials shifted by one)
    "tiny-v3": (128, 384, 6, 4, 384, 6, 4),
}


PEAK_STRIDE = 6          # audio frames (of 20 ms) per decoder position of the "peaked" variant: 0.12 s per token


def position_codes(t, channels, max_timescale=10000):
    """sinusoids() at arbitrary (fractional) positions `t`: (len(t), channels)."""
    log_inc = np.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-log_inc * torch.arange(channels // 2))
    ang = torch.as_tensor(t, dtype=torch.float32)[:, None] * inv[None, :]
    return torch.cat([torch.sin(ang), torch.cos(ang)], dim=1)


def sharpen_cross_attention(model: "Whisper", stride: int = PEAK_STRIDE, gain: float = 1.25, damp: float = 0.05):
    """Turn a random-init model into one whose cross-attention looks like a TRAINED model's on the alignment heads: the
    QK logits of decoder position p have a ridge at audio frame p * stride (monotone in p, a few frames wide, ~12 logits
    above the side lobes) instead of the nearly flat rows random weights give.  Construction (weights only -- the
    architecture and every code path of the reference and of the product are untouched):
      * both residual streams are kept close to their position codes: the encoder's second convolution and every block's
        output projections are damped, so ln_post(x)[f] ~ the sinusoidal code of frame f; the decoder's learned positional
        embedding is set to 2 x the SAME code family evaluated at frame p * stride;
      * every head of the model's ``alignment_heads`` mask (and of the reference's table for the model size) gets query /
        key projections that pick 32 (sin, cos) channel pairs of that code: q . k = sum_k cos(w_k (p * stride - f)).
    Scripted transcripts whose timestamps follow the same p * stride rule (many_helper.peaked_window) then have their
    ridge inside each segment's frame window, like speech."""
    dims = model.dims
    D = dims.n_text_state
    assert dims.n_audio_state == D
    half = D // 2
    with torch.no_grad():
        enc, dec = model.encoder, model.decoder
        enc.conv2.weight.mul_(damp * 0.4)
        enc.conv2.bias.mul_(damp * 0.4)
        for blk in list(enc.blocks) + list(dec.blocks):
            for lin in (blk.attn.out, blk.mlp[2]) + ((blk.cross_attn.out,) if blk.cross_attn is not None else ()):
                lin.weight.mul_(damp)
                lin.bias.mul_(damp)
        for ln in [enc.ln_post] + [blk.cross_attn_ln for blk in dec.blocks]:
            ln.weight.fill_(1.0)
            ln.bias.zero_()
        dev = dec.positional_embedding.device
        dec.positional_embedding.copy_(2.0 * position_codes(torch.arange(dims.n_text_ctx) * float(stride), D).to(dev))
        heads = set()
        if hasattr(model, "alignment_heads"):
            heads |= {(int(l), int(h)) for l, h in model.alignment_heads.to_dense().nonzero().tolist()}
        heads |= {(l, h) for l in range(dims.n_text_layer) for h in range(dims.n_text_head)} if dims.n_text_layer <= 4 else set()
        from_table = {6: [(3, 1), (4, 2), (4, 3), (4, 7), (5, 1), (5, 2), (5, 4), (5, 6)],
                      12: [(5, 3), (5, 9), (8, 0), (8, 4), (8, 7), (8, 8), (9, 0), (9, 7), (9, 9), (10, 5)]}
        heads |= set(from_table.get(dims.n_text_layer, []))
        d_head = D // dims.n_text_head
        pairs = d_head // 2
        for l, h in sorted(heads):
            ca = dec.blocks[l].cross_attn
            ks = (torch.linspace(0.07, 0.67, pairs) * (half - 1)).round().long() + (h % 3)
            sel = torch.zeros(d_head, D)
            sel[torch.arange(pairs), ks] = gain
            sel[pairs + torch.arange(pairs), half + ks] = gain
            ca.query.weight[h * d_head:(h + 1) * d_head] = sel.to(ca.query.weight.device)
            ca.query.bias[h * d_head:(h + 1) * d_head] = 0.0
            ca.key.weight[h * d_head:(h + 1) * d_head] = sel.to(ca.key.weight.device)
    return model


def build_model(name: str = "tiny", seed: int = 0, device="cpu", text_layers=None, audio_layers=None, attention="flat") -> Whisper:
    """Random-initialised Whisper of the named size (``name`` may end in ``.en``).

    The init is chosen so that scripted decoding is well-conditioned: text-token
    embeddings have std 0.25 (peaked next-token distributions: max text
    log-prob well above the log-sum-exp of the timestamp tokens), timestamp
    embeddings std 0.04 (the "timestamps dominate" rule of ApplyTimestampRules
    never fires by accident), attention projections std ~1/sqrt(d) * 2 so that
    cross-attention is not flat.  ``attention="peaked"``: see sharpen_cross_attention (a monotone ridge on the
    alignment heads, as a trained model has; "flat" = plain random init, rows nearly flat after the softmax).
    """
    english = name.endswith(".en")
    base = name[:-3] if english else name
    m, a_s, a_h, a_l, t_s, t_h, t_l = _DIMS[base]
    n_vocab = 51864 if english else (51866 if base in ("large-v3", "tiny-v3") else 51865)
    dims = ModelDimensions(m, 1500, a_s, a_h, audio_layers or a_l, n_vocab, 448, t_s, t_h, text_layers or t_l)
    g = torch.Generator().manual_seed(seed)
    model = Whisper(dims)
    with torch.no_grad():
        for pname, p in model.named_parameters():
            if p.dim() >= 2:
                std = 2.0 / np.sqrt(p.shape[-1] if p.dim() == 2 else p.shape[1] * p.shape[2])
                p.copy_(torch.randn(p.shape, generator=g) * std)
            elif pname.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
        emb = model.decoder.token_embedding.weight
        emb.copy_(torch.randn(emb.shape, generator=g) * 0.25)
        ts0 = n_vocab - 1501
        emb[ts0:] *= 0.16
        model.decoder.positional_embedding.copy_(torch.randn(448, t_s, generator=g) * 0.1)
    assert attention in ("flat", "peaked"), attention
    if attention == "peaked":
        sharpen_cross_attention(model)
    return model.to(device).eval()
