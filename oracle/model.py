"""CPU restatement of train/model.py and train/rl_model.py (Encoder, Decoder, VideoVAE).

Test infrastructure only.  Stochastic ops take caller-supplied noise
(``noise`` dict) because JAX threefry streams cannot be reproduced
(SURVEY.md A.12): keys ``gumbel_u`` (uniform, (b,t,1)), ``reparam_eps``
(normal, like log_variance), ``bernoulli_u`` (uniform, (2b,t,1,1)).
"""
from dataclasses import dataclass

import torch
from einops import rearrange

from . import nn as O
from . import layers as L
from . import unet as U
from .unet import sub


@dataclass
class VAEConfig:
    height: int
    width: int
    channels: int
    patch_size: int
    encoder_depth: int
    decoder_depth: int
    mlp_dim: int
    num_heads: int
    qkv_features: int
    max_temporal_len: int
    spatial_compression_rate: int
    unembedding_upsample_rate: int

    @property
    def last_dim(self):
        return self.channels * self.patch_size * self.patch_size

    @property
    def hw(self):
        return self.height // self.patch_size * self.width // self.patch_size

    @property
    def latent_dim(self):
        return self.last_dim // self.spatial_compression_rate


def encoder_heads(p, x, gumbel_u=None, train=True, flavour="model", dtype=O.F32):
    """What Encoder.__call__ does behind its last block: the mean / variance heads and the two selection layers.
    model.py:53-59 (Gumbel-STE gate) | rl_model.py:53-59 (sigmoid probability).  ``x`` (b, t, hw, c) -> (mean, log_variance, selection)."""
    mean = O.linear(x, p["spatial_compression.kernel"], p["spatial_compression.bias"], dtype)
    variance = O.softplus(O.linear(x, p["variance_estimator.kernel"], p["variance_estimator.bias"], dtype))
    log_variance = O.q(torch.log(O.q(variance, dtype)), dtype)
    si = O.linear(mean, p["selection_layer1.kernel"], p["selection_layer1.bias"], dtype)
    si = rearrange(si, "b t hw 1 -> b t hw")
    logits = O.q(O.linear(si, p["selection_layer2.kernel"], p["selection_layer2.bias"], dtype) + 1, dtype)
    if flavour == "model":
        sel = L.gumbel_sigmoid_ste(logits, gumbel_u, 1.0, train)          # model.py:58
        sel = rearrange(sel, "b t 1 -> b t 1 1")                         # model.py:59
    else:
        sel = O.q(torch.sigmoid(logits), dtype)                          # rl_model.py:59
    return mean, log_variance, sel


def encoder(p, cfg, x, mask, gumbel_u=None, train=True, flavour="model", dtype=O.F32):
    """Encoder.__call__.  model.py:49-60 (Gumbel-STE gate) | rl_model.py:50-60 (sigmoid prob)."""
    x = L.patch_embedding(sub(p, "patch_embedding"), x, cfg.patch_size, dtype)
    for i in range(cfg.encoder_depth):
        x = L.factored_attention(sub(p, f"layers.{i}"), x, mask, cfg.num_heads,
                                 cfg.max_temporal_len, cfg.hw, dtype)
    return encoder_heads(p, x, gumbel_u, train, flavour, dtype)


def decoder(p, cfg, x, mask, dtype=O.F32):
    """Decoder.__call__: Linear -> N x FactoredAttention -> un-patchify -> coarse + UNet(feat).  model.py:90-97."""
    x = O.linear(x, p["spatial_decompression.kernel"], p["spatial_decompression.bias"], dtype)
    for i in range(cfg.decoder_depth):
        x = L.factored_attention(sub(p, f"layers.{i}"), x, mask, cfg.num_heads,
                                 cfg.max_temporal_len, cfg.hw, dtype)
    feat, coarse = L.patch_unembedding(sub(p, "patch_unembedding"), x, cfg.height, cfg.width,
                                       cfg.patch_size, cfg.unembedding_upsample_rate, dtype)
    return O.q(coarse + U.unet(sub(p, "unet"), feat, dtype), dtype)


def reparameterise(mean, log_variance, eps, train=True, dtype=O.F32):
    """z = mean + eps * exp(log_var / 2) if train else mean.  model.py:124-131.

    Mixed precision: ``std = jnp.exp(log_variance / 2)`` is an array of the compute dtype (log_variance is; the halving is exact),
    ``noise`` is jax.random.normal's default float32, so ``noise * std`` and the sum promote to float32 (model.py:124-126)."""
    if not train:
        return mean
    return mean + eps * O.q(torch.exp(log_variance / 2), dtype)


def latent_gate(fill_token, selection, z):
    """compressed_representation = fill_token * (1 - selection) + sampled_latent * selection.  model.py:133 | rl_model.py:144."""
    return fill_token * (1 - selection) + z * selection


def bernoulli_mask(selection, u):
    """jax.random.bernoulli(key, p=selection) with the uniform draw handed in: u < p.  rl_model.py:141-142."""
    return u < selection


def video_vae(p, cfg, x, mask, noise, train=True, dtype=O.F32):
    """VideoVAE.__call__ of train/model.py:119-136 -> 5-tuple."""
    mean, logvar, sel = encoder(sub(p, "encoder"), cfg, x, mask, noise.get("gumbel_u"), train, "model", dtype)
    z = reparameterise(mean, logvar, noise.get("reparam_eps"), train, dtype)
    comp = latent_gate(p["fill_token"], sel, z)                           # model.py:133
    recon = decoder(sub(p, "decoder"), cfg, comp, mask, dtype)
    return recon, comp, sel, logvar, mean


def video_vae_rl(p, cfg, x, mask, noise, train=True, dtype=O.F32):
    """VideoVAE.__call__ of train/rl_model.py:119-147 -> 6-tuple (pair-doubled batch)."""
    mean, logvar, sel = encoder(sub(p, "encoder"), cfg, x, mask, None, train, "rl", dtype)
    z = reparameterise(mean, logvar, noise.get("reparam_eps"), train, dtype)
    sel = rearrange(sel, "b t 1 -> b t 1 1").repeat_interleave(2, dim=0)  # :136
    z = z.repeat_interleave(2, dim=0)                                    # :137
    mean = mean.repeat_interleave(2, dim=0)
    logvar = logvar.repeat_interleave(2, dim=0)
    mask = mask.repeat_interleave(2, dim=0)                              # :140
    sel_mask = bernoulli_mask(sel, noise["bernoulli_u"]).to(z.dtype)     # :142 bernoulli(p=selection)
    comp = latent_gate(p["fill_token"], sel_mask, z)                     # :144
    recon = decoder(sub(p, "decoder"), cfg, comp, mask, dtype)
    return recon, comp, sel, sel_mask, logvar, mean


def init_video_vae(cfg, seed=2, zero_final=True):
    """Parameter tree of VideoVAE.__init__ (model.py:102-115) with Flax default inits."""
    gen = torch.Generator().manual_seed(seed)
    p = {}
    d, ld = cfg.last_dim, cfg.latent_dim
    p["fill_token"] = torch.randn((1, 1, 1, ld), generator=gen) * 0.02
    L.init_linear(p, "encoder.patch_embedding.linear", d, d, gen)
    L.init_ln(p, "encoder.patch_embedding.norm", d)
    L.init_linear(p, "encoder.spatial_compression", d, ld, gen)
    L.init_linear(p, "encoder.variance_estimator", d, ld, gen)
    L.init_linear(p, "encoder.selection_layer1", ld, 1, gen)
    L.init_linear(p, "encoder.selection_layer2", cfg.hw, 1, gen)
    for i in range(cfg.encoder_depth):
        L.init_factored_attention(p, f"encoder.layers.{i}", cfg.mlp_dim, d, cfg.num_heads, cfg.qkv_features, gen)
    u = cfg.unembedding_upsample_rate
    L.init_linear(p, "decoder.patch_unembedding.upsample", d, d * u, gen)
    L.init_linear(p, "decoder.patch_unembedding.downsample", cfg.channels * u, cfg.channels, gen)
    L.init_linear(p, "decoder.patch_unembedding.linear", d, d, gen)
    L.init_linear(p, "decoder.spatial_decompression", ld, d, gen)
    for i in range(cfg.decoder_depth):
        L.init_factored_attention(p, f"decoder.layers.{i}", cfg.mlp_dim, d, cfg.num_heads, cfg.qkv_features, gen)
    up = U.init_unet(cfg.channels * u, 16, 3, cfg.channels, seed=seed + 1000, zero_final=zero_final)
    for k, v in up.items():
        p[f"decoder.unet.{k}"] = v
    return p
