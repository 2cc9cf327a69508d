"""CPU: the oracle against its committed golden vectors, plus self-consistency guards for every Flax semantic that a
PyTorch restatement silently gets wrong.  The reference holds no golden vectors
("parity unpinned"), so these are the pins the oracle has."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import nn as O, unet as OU, layers as OL, model as OM, loss as OLoss, optim as OOpt
from util import assert_close, assert_close_scaled, rnd

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(G, name)).items()}


def test_golden_unet_small():
    d = load("unet_small.npz")
    p = {k[2:]: v.clone().requires_grad_(True) for k, v in d.items() if k.startswith("p.")}
    x = d["x"].clone().requires_grad_(True)
    y = OU.unet(p, x)
    y.backward(d["gy"])
    # tolerances cover oneDNN summation-order differences between thread counts, nothing more
    assert_close(y, d["y"], rtol=1e-4, atol=1e-5, what="y")
    assert_close_scaled(x.grad, d["dx"], rel=1e-4, what="dx")
    for k, v in d.items():
        if k.startswith("g."):
            assert_close_scaled(p[k[2:]].grad, v, rel=1e-4, what=k)


@pytest.mark.parametrize("t", [4, 16])
def test_golden_temporal_attention(t):
    d = load(f"temporal_attn_T{t}.npz")
    a, heads, dd = d["qkv"].shape[0], 2, 16
    q, k, v = torch.chunk(d["qkv"], 3, dim=-1)
    sp = lambda z: z.reshape(a, t, heads, dd)
    cos, sin = OL.rope_tables(dd, 64)
    qr, kr = OL.rope(O.layer_norm(sp(q), d["q_scale"], None), O.layer_norm(sp(k), d["k_scale"], None), cos, sin)
    out = OL.dot_product_attention(qr, kr, sp(v), d["mask"].bool().reshape(a, 1, 1, t)).reshape(a, t, -1)
    assert_close(out, d["out"], rtol=1e-5, atol=1e-6)


def test_golden_reparam_kl_loss():
    d = load("reparam_kl_loss.npz")
    assert_close(OM.reparameterise(d["mean"], d["logvar"], d["eps"]), d["z"], rtol=1e-6, atol=1e-6)
    assert_close(OLoss.kl_per_sample(d["mean"], d["logvar"], d["mask"]), d["kl"], rtol=1e-5, atol=1e-7)
    mse, mae = OLoss.masked_mse_mae(d["video"], d["recon"], d["mask"])
    assert_close(mse, d["mse"], rtol=1e-5, atol=1e-8)
    assert_close(mae, d["mae"], rtol=1e-5, atol=1e-8)


def test_golden_tiny_vae():
    d = load("tiny_vae.npz")
    cfg = OM.VAEConfig(32, 32, 3, 8, 1, 1, 64, 4, 32, 8, 4, 4)
    p = OM.init_video_vae(cfg, seed=3, zero_final=False)
    noise = {k[6:]: v for k, v in d.items() if k.startswith("noise.")}
    em = OLoss.expand_mask(d["mask"].bool(), cfg.hw)
    outs = OM.video_vae(p, cfg, d["video"], em, noise)
    loss, aux = OLoss.loss_fn_plain(outs, d["video"], d["mask"])
    assert_close(outs[0], d["model.recon"], rtol=1e-4, atol=1e-5, what="recon")
    assert_close(loss, d["model.loss"], rtol=1e-5, atol=1e-6, what="loss")
    outs = OM.video_vae_rl(p, cfg, d["video"], em, noise)
    loss, aux = OLoss.loss_fn_rl(outs, d["video"], d["mask"])
    assert_close(loss, d["rl.loss"], rtol=1e-5, atol=1e-6, what="rl loss")
    assert outs[0].shape == (4, 8, 32, 32, 3)
    assert set(outs[3].unique().tolist()) <= {0.0, 1.0}                  # selection_mask is binary


# ---------------------------------------------------------------------------------------------- semantic guards
def test_conv_transpose_is_unflipped_dilated_correlation():
    """A.4: nnx.ConvTranspose(transpose_kernel=False) = dilate lhs by stride, pad (1,1), correlate without flipping."""
    x, k, b = rnd((1, 2, 3, 4, 5), 1), rnd((1, 2, 2, 5, 6), 2), rnd((6,), 3)
    assert_close(O.conv_transpose_1x2x2(x, k, b), O.conv_transpose_1x2x2_explicit(x, k, b), rtol=1e-5, atol=1e-6)
    # and it is NOT torch's ConvTranspose3d with the same kernel order (which would be the flipped one)
    w = k.permute(3, 4, 0, 1, 2).contiguous()
    torch_ct = F.conv_transpose3d(x.permute(0, 4, 1, 2, 3), w, stride=(1, 2, 2)).permute(0, 2, 3, 4, 1) + b
    assert float((torch_ct - O.conv_transpose_1x2x2(x, k, b)).abs().max()) > 1e-2
    # every input voxel owns a disjoint 2x2 output block: out[2i+d] = x[i] K[1-d]
    y = O.conv_transpose_1x2x2(x, k, torch.zeros(6))
    assert_close(y[0, 0, 2 * 1 + 0, 2 * 2 + 1], x[0, 0, 1, 2] @ k[0, 1, 0], rtol=1e-5, atol=1e-6)


def test_group_norm_matches_torch_with_flax_eps_and_spans_time():
    """A.3: same reduction set as torch GroupNorm on (N,C,D,H,W) but eps=1e-6; stats span (t,h,w)."""
    x = rnd((2, 3, 4, 5, 16), 4) * 2 + 0.5
    sc, bi = 1 + 0.1 * rnd((16,), 5), 0.1 * rnd((16,), 6)
    want = F.group_norm(x.permute(0, 4, 1, 2, 3), 8, sc, bi, eps=1e-6).permute(0, 2, 3, 4, 1)
    assert_close(O.group_norm(x, sc, bi, 8), want, rtol=1e-4, atol=1e-5)
    x2 = x.clone()
    x2[:, 2] += 3.0
    assert float((O.group_norm(x2, sc, bi, 8)[:, 0] - O.group_norm(x, sc, bi, 8)[:, 0]).abs().max()) > 1e-3


def test_conv_same_is_cross_correlation_with_flax_kernel_layout():
    """A.1: kernel (kt,kh,kw,Cin,Cout), no flip, zero pad (k-1)/2: check one output voxel by hand."""
    x, k, b = rnd((1, 3, 5, 5, 2), 7), rnd((3, 3, 3, 2, 4), 8), rnd((4,), 9)
    y = O.conv3d_same(x, k, b)
    acc = b.clone()
    for a in range(3):
        for bb in range(3):
            for c in range(3):
                t, h, w = 1 + a - 1, 0 + bb - 1, 4 + c - 1
                if 0 <= t < 3 and 0 <= h < 5 and 0 <= w < 5:
                    acc = acc + x[0, t, h, w] @ k[a, bb, c]
    assert_close(y[0, 1, 0, 4], acc, rtol=1e-5, atol=1e-6)


def test_layer_norm_fast_variance_and_eps():
    x = rnd((4, 7, 32), 10) * 3 + 1
    sc, bi = 1 + 0.1 * rnd((32,), 11), 0.1 * rnd((32,), 12)
    assert_close(O.layer_norm(x, sc, bi), F.layer_norm(x, (32,), sc, bi, eps=1e-6), rtol=1e-4, atol=1e-5)


def test_attention_matches_torch_sdpa_and_masked_equals_truncated():
    """A.7 + reference property train/scratch.py:46-57."""
    q, k, v = rnd((2, 12, 3, 16), 13), rnd((2, 12, 3, 16), 14), rnd((2, 12, 3, 16), 15)
    want = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)).transpose(1, 2)
    assert_close(OL.dot_product_attention(q, k, v, None), want, rtol=1e-4, atol=1e-5)
    L = 5
    mask = (torch.arange(12) < L).reshape(1, 1, 1, 12).expand(2, 1, 1, 12)
    full = OL.dot_product_attention(q, k, v, mask)
    trunc = OL.dot_product_attention(q[:, :L], k[:, :L], v[:, :L], None)
    assert_close(full[:, :L], trunc, rtol=1e-5, atol=1e-6)


def test_rope_is_a_rotation_and_relative():
    cos, sin = OL.rope_tables(16, 32)
    q, k = rnd((1, 8, 2, 16), 16), rnd((1, 8, 2, 16), 17)
    qr, kr = OL.rope(q, k, cos, sin)
    assert_close(qr.norm(dim=-1), q.norm(dim=-1), rtol=1e-5, atol=1e-6)            # norm preserving
    # scores depend on relative position only: shift both by 3 frames
    q2 = torch.cat([torch.zeros(1, 3, 2, 16), q], 1)
    k2 = torch.cat([torch.zeros(1, 3, 2, 16), k], 1)
    qr2, kr2 = OL.rope(q2, k2, cos, sin)
    s1 = torch.einsum("btnh,bsnh->bnts", qr, kr)
    s2 = torch.einsum("btnh,bsnh->bnts", qr2[:, 3:], kr2[:, 3:])
    assert_close(s1, s2, rtol=1e-4, atol=1e-4)


def test_round_ste_identity_gradient_and_gumbel_binary():
    """claude_distributed/test_rl_model.py:178-188."""
    x = rnd((5,), 18).requires_grad_(True)
    OL.round_ste(x * 3).sum().backward()
    assert_close(x.grad, torch.full((5,), 3.0))
    u = torch.rand(4, 6, 1, generator=torch.Generator().manual_seed(1))
    g = OL.gumbel_sigmoid_ste(rnd((4, 6, 1), 19), u)
    assert set(g.unique().tolist()) <= {0.0, 1.0}


def test_batch_isolation():
    """train/human_tests.py:90-96: sample 0 alone == sample 0 in a batch (GroupNorm is per-sample)."""
    p = OU.init_unet(4, 8, 1, 3, seed=2, zero_final=False)
    x = rnd((2, 2, 8, 8, 4), 20)
    assert_close(OU.unet(p, x)[:1], OU.unet(p, x[:1]), rtol=1e-4, atol=1e-5)


def test_optax_schedule_and_clip_and_adam():
    """A.13: lr(0)=0, peak at warmup end, end value after decay; clip only when ||g|| >= c; adam == torch.optim.Adam."""
    kw = dict(init_value=0.0, peak_value=2e-5, warmup_steps=20000 // math.sqrt(2), decay_steps=1_000_000, end_value=2e-6)
    assert kw["warmup_steps"] == 14142.0
    assert OOpt.warmup_cosine_decay_schedule(0, **kw) == 0.0
    assert abs(OOpt.warmup_cosine_decay_schedule(14142, **kw) - 2e-5) < 1e-12
    assert abs(OOpt.warmup_cosine_decay_schedule(2_000_000, **kw) - 2e-6) < 1e-15
    g = {"a": torch.tensor([0.3, 0.4])}
    c, n = OOpt.clip_by_global_norm(g, 1.0)
    assert torch.equal(c["a"], g["a"]) and abs(float(n) - 0.5) < 1e-6
    c, n = OOpt.clip_by_global_norm({"a": torch.tensor([3.0, 4.0])}, 1.0)
    assert_close(c["a"], torch.tensor([0.6, 0.8]), rtol=1e-6, atol=1e-7)
    w = torch.nn.Parameter(rnd((7,), 21))
    topt = torch.optim.Adam([w], lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
    adam = OOpt.Adam({"w": w.detach().clone()})
    po = {"w": w.detach().clone()}
    for s in range(3):
        gr = rnd((7,), 22 + s)
        w.grad = gr.clone()
        topt.step()
        po = adam.update(po, {"w": gr}, 1e-2)
        assert_close(po["w"], w.detach(), rtol=1e-5, atol=1e-7)


def test_loss_pair_statistics():
    """jnp.std is the population std; pairs are (2k, 2k+1) (A.14, rl_nonadversarial.py:150-153)."""
    per = torch.tensor([1.0, 3.0, 2.0, 2.0])
    pairs = per.reshape(2, 2)
    std = pairs.std(dim=1, unbiased=False)
    assert_close(std, torch.tensor([1.0, 0.0]))
    assert torch.equal(torch.arange(4).repeat_interleave(2, 0)[:4], torch.tensor([0, 0, 1, 1]))
