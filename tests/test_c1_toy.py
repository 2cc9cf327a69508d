"""Config C1 (BASELINE.json configs[0]): the toy two-ConvBlock3D encoder / decoder VAE on (1, 8, 64, 64, 3) clips.

CPU half (runs everywhere): the oracle's toy obeys the reference's property tests for a trainable model -- shape contract, finite
non-zero gradients, loss decreases over 10 steps at lr 1e-3 on a fixed batch (claude_distributed/test_training_loop.py:168-200).
GPU half (-m gpu): the HIP product (video_vae_amd/toy.py) against that oracle -- reconstruction, loss terms and EVERY parameter
gradient at the fp32 bar of BASELINE.json (rtol 1e-3 / atol 1e-4), and the 10-step recon + KL loss curve.
"""
import pytest
import torch

from oracle import optim as OOpt
from oracle import toy as OT
from util import assert_close, assert_close_scaled, grad_floor

SHAPE = (1, 8, 64, 64, 3)
LR = 1e-3


def _case(seed=0):
    g = torch.Generator().manual_seed(seed)
    video = torch.rand(SHAPE, generator=g)
    mask = torch.ones(SHAPE[:2])
    mask[0, 6:] = 0                                                      # ragged clip: the last two frames are padding
    eps = [torch.randn(SHAPE[:-1] + (8,), generator=g) for _ in range(10)]
    return OT.init_toy(seed=4), video, mask, eps


def _oracle_step(p, adam, video, mask, eps):
    pr = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    loss, aux = OT.toy_loss(OT.toy_vae(pr, video, eps), video, mask)
    loss.backward()
    grads = {k: v.grad for k, v in pr.items()}
    clipped, _ = OOpt.clip_by_global_norm(grads, 1.0)
    return adam.update({k: v.detach() for k, v in pr.items()}, clipped, LR), loss.detach(), aux, grads


def test_oracle_toy_contract_and_learning():
    p, video, mask, eps = _case()
    recon, z, lv, mean = OT.toy_vae(p, video, eps[0])
    assert recon.shape == SHAPE and z.shape == SHAPE[:-1] + (8,) and lv.shape == mean.shape == z.shape
    assert torch.equal(OT.toy_vae(p, video, eps[0], train=False)[1], mean)           # eval: z = mean (model.py:129-131)
    adam = OOpt.Adam(p)
    losses = []
    for i in range(10):
        p, loss, _aux, grads = _oracle_step(p, adam, video, mask, eps[i])
        assert all(torch.isfinite(g).all() for g in grads.values())
        if i == 0:
            assert all(float(g.abs().max()) > 0 for k, g in grads.items() if not k.endswith("conv.bias"))
        losses.append(float(loss))
    assert losses[-1] < losses[0], losses
    # masked frames do not reach the loss: changing them changes nothing
    v2 = video.clone()
    v2[0, 6:] = 0.123
    a = OT.toy_loss(OT.toy_vae(p, video, eps[0]), video, mask)[1]["MSE"]
    r = OT.toy_vae(p, video, eps[0])
    b = OT.toy_loss(r, v2, mask)[1]["MSE"]
    assert torch.equal(a, b)


@pytest.mark.gpu
def test_c1_toy_fwd_bwd_and_loss_curve_vs_oracle(dev):
    import video_vae_amd as V
    from video_vae_amd import optim, toy
    p, video, mask, eps = _case()
    m = toy.ToyVAE(3, 16, 8, V.Rngs(0), dtype=torch.float32)
    sd = m.state_dict()
    assert set(sd) == set(p), set(sd) ^ set(p)
    with torch.no_grad():
        for k, v in p.items():
            sd[k].copy_(v)
    m = m.to(dev)
    vg, mg = video.to(dev), mask.to(dev)
    # ---- one forward + backward: output, loss terms, every gradient
    rngs = V.Rngs(1)
    rngs.inject("reparam_eps", eps[0])
    loss_g, aux_g = toy.toy_loss_fn(m, vg, mg, rngs)
    loss_g.backward()
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    loss_o, aux_o = OT.toy_loss(OT.toy_vae(po, video, eps[0]), video, mask)
    loss_o.backward()
    assert_close(aux_g["reconstruction"], aux_o["reconstruction"], what="C1 reconstruction")          # rtol 1e-3 / atol 1e-4
    for k in ("MSE", "kl_loss"):
        assert_close(aux_g[k], aux_o[k], rtol=1e-3, atol=1e-6, what=k)
    assert_close(loss_g, loss_o, rtol=1e-3, atol=1e-6, what="loss")
    ref = {k: v.grad for k, v in po.items()}
    for k, prm in m.named_parameters():
        assert_close_scaled(prm.grad, ref[k], rel=1e-3, what=f"d{k}", floor=grad_floor(k, ref))
    # ---- 10 optimizer steps from the same state: recon + KL loss curve (north_star: "loss curve matching the CPU reference")
    with torch.no_grad():
        for k, v in p.items():
            sd[k].copy_(v)
    opt = optim.Optimizer(m, LR, bf16_shadow=False)
    adam = OOpt.Adam(p)
    for i in range(10):
        rngs.inject("reparam_eps", eps[i])
        lg, ag = toy.toy_train_step(m, opt, vg, mg, rngs)
        p, lo, ao, _ = _oracle_step(p, adam, video, mask, eps[i])
        assert_close(lg, lo, rtol=1e-3, atol=1e-4, what=f"loss, step {i}")
        assert_close(ag["MSE"], ao["MSE"], rtol=1e-3, atol=1e-4, what=f"MSE, step {i}")
        assert_close(ag["kl_loss"], ao["kl_loss"], rtol=1e-3, atol=1e-4, what=f"KL, step {i}")
    for k, prm in m.named_parameters():
        # a conv bias in front of a one-channel-per-group GroupNorm (dec2: 3 channels, 3 groups) has a gradient that is zero in exact
        # arithmetic; Adam normalises its rounding noise into steps of +-lr, a different walk on either side (measured: 7.6e-3 apart after
        # 10 steps of 1e-3).  It cannot reach the output (the norm subtracts it), which the matching loss curve above shows: not compared.
        if k.endswith("conv.bias") and prm.numel() <= 8:
            continue
        assert_close_scaled(prm.detach(), p[k], rel=1e-3, what=f"{k} after 10 steps")
