"""CPU restatement of train/layers.py (patch (un)embedding, RoPE, attention, MLP,
FactoredAttention, round_ste, GumbelSigmoidSTE).  Test infrastructure only."""
import math

import torch
from einops import rearrange

from . import nn as O
from .unet import sub


def patch_embedding(p, x, patch_size, dtype=O.F32):
    """PatchEmbedding.__call__.  layers.py:20-27."""
    x = rearrange(x, "b t (h p1) (w p2) c -> b t (h w) (p1 p2 c)", p1=patch_size, p2=patch_size)
    x = O.q(x, dtype)
    x = O.layer_norm(x, p["norm.scale"], p["norm.bias"], dtype)
    return O.linear(x, p["linear.kernel"], p["linear.bias"], dtype)


def patch_unembedding(p, x, height, width, patch_size, upsample_rate, dtype=O.F32):
    """PatchUnEmbedding.__call__ -> (conv features (b,t,H,W,c*u), coarse (b,t,H,W,c)).  layers.py:45-55."""
    x = O.linear(x, p["linear.kernel"], p["linear.bias"], dtype)
    x = O.linear(x, p["upsample.kernel"], p["upsample.bias"], dtype)
    feat = rearrange(x, "b t (h w) (p1 p2 c u) -> b t (h p1) (w p2) (c u)",
                     p1=patch_size, p2=patch_size, h=height // patch_size, w=width // patch_size,
                     u=upsample_rate)
    coarse = O.linear(feat, p["downsample.kernel"], p["downsample.bias"], dtype)
    return feat, coarse


def rope_tables(head_dim, max_len, alpha=1.0, base=10000.0):
    """RotaryEmbedding.__init__ cos/sin caches, shape (max_len, head_dim).  layers.py:86-103."""
    ntk_base = base * (alpha ** (head_dim / (head_dim - 2)))
    inv_freq = 1.0 / (ntk_base ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    t = torch.arange(max_len, dtype=torch.float32)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return torch.cos(emb), torch.sin(emb)


def rotate_half(x):
    """layers.py:80-83."""
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def rope(q, k, cos, sin, dtype=O.F32):
    """rotate_queries_and_keys on (b, seq, heads, hd); tables cast to q dtype.  layers.py:105-129."""
    s = q.shape[1]
    c = O.q(cos[:s], dtype)[None, :, None, :]
    sn = O.q(sin[:s], dtype)[None, :, None, :]
    qr = O.q(O.q(q * c, dtype) + O.q(rotate_half(q) * sn, dtype), dtype)
    kr = O.q(O.q(k * c, dtype) + O.q(rotate_half(k) * sn, dtype), dtype)
    return qr, kr


def dot_product_attention(q, k, v, mask, dtype=O.F32):
    """jax.nn.dot_product_attention, (B,T,N,H) inputs, bool mask True=attend.

    logits in fp32, scaled 1/sqrt(H), masked to a large negative, softmax fp32,
    probabilities cast to the value dtype.  layers.py:168.
    """
    hd = q.shape[-1]
    logits = torch.einsum("btnh,bsnh->bnts", q, k) * (1.0 / math.sqrt(hd))
    if mask is not None:
        big_neg = -0.7 * torch.finfo(torch.float32).max
        logits = torch.where(mask.to(torch.bool), logits, torch.full((), big_neg))
    probs = O.q(torch.softmax(logits, dim=-1), dtype)
    return O.q(torch.einsum("bnts,bsnh->btnh", probs, v), dtype)


def attention(p, x, num_heads, max_len, mask=None, dtype=O.F32):
    """Attention.__call__: LN -> QKV -> per-head LN(q), LN(k) (no bias) -> RoPE -> SDPA -> out.  layers.py:158-171."""
    x = O.layer_norm(x, p["input_norm.scale"], p["input_norm.bias"], dtype)
    qkv = O.linear(x, p["qkv_projection.kernel"], p["qkv_projection.bias"], dtype)
    q, k, v = torch.chunk(qkv, 3, dim=-1)
    q = rearrange(q, "b s (h d) -> b s h d", h=num_heads)
    k = rearrange(k, "b s (h d) -> b s h d", h=num_heads)
    v = rearrange(v, "b s (h d) -> b s h d", h=num_heads)
    q = O.layer_norm(q, p["q_norm.scale"], None, dtype)
    k = O.layer_norm(k, p["k_norm.scale"], None, dtype)
    cos, sin = rope_tables(q.shape[-1], max_len)
    q, k = rope(q, k, cos, sin, dtype)
    o = dot_product_attention(q, k, v, mask, dtype)
    o = rearrange(o, "b s h d -> b s (h d)")
    return O.linear(o, p["out_projection.kernel"], p["out_projection.bias"], dtype)


def mlp(p, x, dtype=O.F32):
    """MLP.__call__: LN -> Linear -> SiLU -> Linear.  layers.py:191-196."""
    x = O.layer_norm(x, p["norm.scale"], p["norm.bias"], dtype)
    x = O.linear(x, p["linear1.kernel"], p["linear1.bias"], dtype)
    x = O.silu(x, dtype)
    return O.linear(x, p["linear2.kernel"], p["linear2.bias"], dtype)


def factored_attention(p, x, temporal_mask, num_heads, max_temporal_len, max_spatial_len, dtype=O.F32):
    """FactoredAttention.__call__ (mask pre-expanded to (b*hw,1,1,t)).  layers.py:209-224."""
    b, t, hw, c = x.shape
    tx = rearrange(x, "b t hw c -> (b hw) t c")
    tx = O.q(tx + attention(sub(p, "TemporalAttention"), tx, num_heads, max_temporal_len, temporal_mask, dtype), dtype)
    tx = O.q(tx + mlp(sub(p, "TemporalMLP"), tx, dtype), dtype)
    x = rearrange(tx, "(b hw) t c -> b t hw c", b=b, hw=hw)
    sx = rearrange(x, "b t hw c -> (b t) hw c")
    sx = O.q(sx + attention(sub(p, "SpatialAttention"), sx, num_heads, max_spatial_len, None, dtype), dtype)
    sx = O.q(sx + mlp(sub(p, "SpatialMLP"), sx, dtype), dtype)
    return rearrange(sx, "(b t) hw c -> b t hw c", b=b, t=t)


class _RoundSTE(torch.autograd.Function):
    """round_ste: round forward, identity backward.  layers.py:226-236."""

    @staticmethod
    def forward(ctx, x):
        return torch.round(x)

    @staticmethod
    def backward(ctx, g):
        return g


def round_ste(x):
    return _RoundSTE.apply(x)


def gumbel_sigmoid_ste(logits, u=None, temperature=1.0, train=True):
    """GumbelSigmoidSTE.__call__ with caller-supplied uniform noise ``u``.  layers.py:242-252."""
    if train:
        eps = 1e-20
        u = torch.clamp(u, eps, 1.0 - eps)
        noise = torch.log(u / (1 - u))
        return round_ste(torch.sigmoid((logits + noise) / temperature))
    return torch.round(torch.sigmoid(logits / temperature))


# ---------------------------------------------------------------- initialisers

def init_linear(p, prefix, fin, fout, gen, scale=1.0):
    p[f"{prefix}.kernel"] = O.lecun_normal_((fin, fout), fin, gen, scale)
    p[f"{prefix}.bias"] = torch.zeros(fout)


def init_ln(p, prefix, n, bias=True):
    p[f"{prefix}.scale"] = torch.ones(n)
    if bias:
        p[f"{prefix}.bias"] = torch.zeros(n)


def init_attention(p, prefix, in_features, num_heads, qkv_features, gen):
    """Attention.__init__.  layers.py:132-156."""
    init_linear(p, f"{prefix}.qkv_projection", in_features, 3 * qkv_features, gen)
    init_linear(p, f"{prefix}.out_projection", qkv_features, in_features, gen, scale=1e-2)
    init_ln(p, f"{prefix}.input_norm", in_features)
    init_ln(p, f"{prefix}.q_norm", qkv_features // num_heads, bias=False)
    init_ln(p, f"{prefix}.k_norm", qkv_features // num_heads, bias=False)


def init_mlp(p, prefix, in_features, mlp_dim, gen):
    """MLP.__init__.  layers.py:175-189."""
    init_ln(p, f"{prefix}.norm", in_features)
    init_linear(p, f"{prefix}.linear1", in_features, mlp_dim, gen)
    init_linear(p, f"{prefix}.linear2", mlp_dim, in_features, gen, scale=1e-2)


def init_factored_attention(p, prefix, mlp_dim, in_features, num_heads, qkv_features, gen):
    """FactoredAttention.__init__.  layers.py:199-207."""
    init_attention(p, f"{prefix}.SpatialAttention", in_features, num_heads, qkv_features, gen)
    init_mlp(p, f"{prefix}.SpatialMLP", in_features, mlp_dim, gen)
    init_attention(p, f"{prefix}.TemporalAttention", in_features, num_heads, qkv_features, gen)
    init_mlp(p, f"{prefix}.TemporalMLP", in_features, mlp_dim, gen)
