#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (run from the repo root: python tests/golden/make_golden.py).

The reference (JAX/Flax) cannot run in this image (jax/flax absent, no network), so these are NOT reference outputs:
they freeze the oracle's own outputs on seeded inputs ("parity unpinned", see oracle/__init__.py).  They guard the
oracle against drift (tests/test_oracle.py, CPU) and give the GPU tests a fixed target that does not depend on the
oracle source at test time (tests/test_gpu_golden.py).  Inputs are stored with the outputs so the files are
self-contained.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import nn as O, unet as OU, layers as OL, model as OM, loss as OLoss  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name), **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v))
                                                    for k, v in arrs.items()})
    print(name, {k: tuple(np.asarray(v.detach() if torch.is_tensor(v) else v).shape) for k, v in arrs.items()})


def unet_case():
    """2-level UNet on (1,4,16,16,12): output, input gradient and three representative parameter gradients."""
    p = OU.init_unet(12, 8, 2, 3, seed=11, zero_final=False)
    x = rnd((1, 4, 16, 16, 12), 12, 0.5).requires_grad_(True)
    gy = rnd((1, 4, 16, 16, 3), 13)
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    y = OU.unet(po, x)
    y.backward(gy)
    keep = ["patch_mixer.kernel", "encoders.0.conv1.norm.scale", "decoders.0.upsample.kernel", "final_conv.kernel"]
    save("unet_small.npz", x=x, gy=gy, y=y, dx=x.grad, **{"p." + k: v for k, v in p.items()},
         **{"g." + k: po[k].grad for k in keep})


def attention_case():
    for t in (4, 16):
        a, heads, d = 3, 2, 16
        qkv = rnd((a, t, 3 * heads * d), 20 + t).requires_grad_(True)
        qs = (1 + 0.2 * rnd((d,), 21)).requires_grad_(True)
        ks = (1 + 0.2 * rnd((d,), 22)).requires_grad_(True)
        go = rnd((a, t, heads * d), 23)
        lens = torch.tensor([t, max(1, t - 1), max(1, t // 2)])
        mask = (torch.arange(t)[None, :] < lens[:, None]).reshape(a, 1, 1, t)
        q, k, v = torch.chunk(qkv, 3, dim=-1)
        sp = lambda z: z.reshape(a, t, heads, d)
        qn = O.layer_norm(sp(q), qs, None)
        kn = O.layer_norm(sp(k), ks, None)
        cos, sin = OL.rope_tables(d, 64)
        qr, kr = OL.rope(qn, kn, cos, sin)
        out = OL.dot_product_attention(qr, kr, sp(v), mask).reshape(a, t, heads * d)
        out.backward(go)
        save(f"temporal_attn_T{t}.npz", qkv=qkv, q_scale=qs, k_scale=ks, mask=mask.reshape(a, t).to(torch.uint8), go=go,
             out=out, dqkv=qkv.grad, dq_scale=qs.grad, dk_scale=ks.grad)


def reparam_kl_case():
    b, t, hw, c = 2, 6, 4, 12
    mean, lv, eps = rnd((b, t, hw, c), 30), 0.5 * rnd((b, t, hw, c), 31) - 1, rnd((b, t, hw, c), 32)
    mask = torch.ones(b, t)
    mask[1, 4:] = 0
    z = OM.reparameterise(mean, lv, eps)
    kl = OLoss.kl_per_sample(mean, lv, mask)
    video = torch.rand((b, t, 8, 8, 3), generator=torch.Generator().manual_seed(33))
    recon = video + 0.2 * rnd((b, t, 8, 8, 3), 34)
    mse, mae = OLoss.masked_mse_mae(video, recon, mask)
    save("reparam_kl_loss.npz", mean=mean, logvar=lv, eps=eps, mask=mask, z=z, kl=kl, video=video, recon=recon, mse=mse, mae=mae)


def vae_case():
    """Tiny VAE (32x32, patch 8, depth 1, as in the reference's claude_distributed/test_distributed.py:113-120 sizes)."""
    cfg = OM.VAEConfig(32, 32, 3, 8, 1, 1, 64, 4, 32, 8, 4, 4)
    p = OM.init_video_vae(cfg, seed=3, zero_final=False)
    b, t = 2, 8
    g = torch.Generator().manual_seed(7)
    video = torch.rand((b, t, 32, 32, 3), generator=torch.Generator().manual_seed(0))
    mask = torch.ones(b, t)
    mask[1, 6:] = 0
    noise = {"gumbel_u": torch.rand((b, t, 1), generator=g), "reparam_eps": torch.randn((b, t, cfg.hw, cfg.latent_dim), generator=g),
             "bernoulli_u": torch.rand((2 * b, t, 1, 1), generator=g)}
    em = OLoss.expand_mask(mask.bool(), cfg.hw)
    out = {}
    for flav, fn, lf in (("model", OM.video_vae, OLoss.loss_fn_plain), ("rl", OM.video_vae_rl, OLoss.loss_fn_rl)):
        outs = fn(p, cfg, video, em, noise)
        loss, aux = lf(outs, video, mask)
        out[f"{flav}.loss"] = loss
        out[f"{flav}.recon"] = outs[0]
        for k, v in aux.items():
            if k != "reconstruction":
                out[f"{flav}.{k}"] = v
    save("tiny_vae.npz", video=video, mask=mask, **{"noise." + k: v for k, v in noise.items()}, **out)


if __name__ == "__main__":
    torch.set_num_threads(4)
    unet_case()
    attention_case()
    reparam_kl_case()
    vae_case()
