"""VGG16 perceptual loss (reference train/vgg_tests.py): oracle self-checks on the CPU, HIP path vs oracle on the GPU."""
import pytest
import torch
import torch.nn.functional as F

from oracle import perceptual as OP
from util import assert_close, assert_close_scaled, rnd


def _params(seed=0, scale=1.0):
    p = {}
    for i, (name, (cin, cout)) in enumerate({"conv1_1": (3, 64), "conv1_2": (64, 64), "conv2_1": (64, 128)}.items()):
        p[f"{name}.weight"] = rnd((3, 3, cin, cout), seed + i, scale / (9 * cin) ** 0.5)
        p[f"{name}.bias"] = rnd((cout,), seed + 10 + i, 0.1)
    return p


def test_oracle_matches_textbook_vgg_head():
    """The restatement against an independently written NCHW torch.nn stack (the published VGG16 layer recipe), and the loss
    variants against each other: scalar = mean of the per-sample one (vgg_tests.py:66 vs :95 with equal-sized frames)."""
    p = _params()
    x, y = torch.rand(2, 3, 16, 16, 3, generator=torch.Generator().manual_seed(1)), torch.rand(2, 3, 16, 16, 3, generator=torch.Generator().manual_seed(2))
    feats = OP.vgg_features(p, x.reshape(6, 16, 16, 3))
    mean, std = torch.tensor(OP.MEAN).view(1, 3, 1, 1), torch.tensor(OP.STD).view(1, 3, 1, 1)
    h = (x.reshape(6, 16, 16, 3).permute(0, 3, 1, 2) - mean) / std
    convs = {}
    for name in ("conv1_1", "conv1_2", "conv2_1"):
        w = p[f"{name}.weight"]
        conv = torch.nn.Conv2d(w.shape[2], w.shape[3], 3, padding=1)
        with torch.no_grad():
            conv.weight.copy_(w.permute(3, 2, 0, 1)); conv.bias.copy_(p[f"{name}.bias"])
        convs[name] = conv
    r11 = F.relu(convs["conv1_1"](h)); r12 = F.relu(convs["conv1_2"](r11)); r21 = F.relu(convs["conv2_1"](F.max_pool2d(r12, 2)))
    for k, v in (("relu1_1", r11), ("relu1_2", r12), ("relu2_1", r21)):
        assert_close(feats[k], v.permute(0, 2, 3, 1), rtol=1e-5, atol=1e-5, what=k)
    assert feats["relu2_1"].shape == (6, 8, 8, 128)
    per = OP.adversarial_perceptual_loss(p, x, y)
    assert per.shape == (2,)
    assert_close(per.mean(), OP.perceptual_loss(p, x, y), rtol=1e-5, atol=1e-7)
    assert float(OP.perceptual_loss(p, x, x)) == 0.0


def test_load_vgg_parameter_files(tmp_path):
    import numpy as np
    from video_vae_amd import perceptual as P
    model, params = P.load_vgg(pretrained=None)
    assert sorted(params) == sorted(f"{n}.{s}" for n in ("conv1_1", "conv1_2", "conv2_1") for s in ("weight", "bias"))
    assert params["conv2_1.weight"].shape == (3, 3, 64, 128) and params["conv1_1.weight"].dtype == torch.float32
    assert torch.equal(params["conv1_2.weight"], params["conv1_2.weight"].bfloat16().float())        # cast like vgg_tests.py:30
    np.savez(tmp_path / "vgg.npz", **{k: v.numpy() for k, v in _params().items()})
    _, loaded = P.load_vgg(pretrained=str(tmp_path / "vgg.npz"), dtype=torch.float32)
    for k, v in _params().items():
        assert torch.equal(loaded[k], v)
    torch.save({k: v for k, v in _params().items() if "conv2" not in k}, tmp_path / "short.pt")
    with pytest.raises(KeyError):
        P.load_vgg(pretrained=str(tmp_path / "short.pt"))
    with pytest.raises(ValueError):
        P.load_vgg(pretrained="imagenet")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_perceptual_loss_vs_oracle(dev, dtype):
    """Features, per-sample loss and the gradient w.r.t. the reconstruction at (2, 4, 32, 32, 3); target_div = 2 equals repeating
    the target (rl_nonadversarial.py:110,125)."""
    from video_vae_amd import perceptual as P
    model = P.VGG16Features(normalize=True, dtype=dtype)
    p = {k: (v.to(dtype).float() if dtype != torch.float32 else v) for k, v in _params().items()}
    pd = {k: v.to(dev) for k, v in p.items()}
    g = torch.Generator().manual_seed(5)
    x = torch.rand(4, 4, 32, 32, 3, generator=g)
    tgt = torch.rand(2, 4, 32, 32, 3, generator=g)
    tgt2 = tgt.repeat_interleave(2, dim=0)

    xr = x.clone().requires_grad_(True)
    ref = OP.adversarial_perceptual_loss(p, xr, tgt2)
    w = torch.tensor([1.0, -0.5, 2.0, 0.25])
    (ref * w).sum().backward()
    feats_ref = OP.vgg_features(p, x.reshape(16, 32, 32, 3))

    xd = x.to(dev).requires_grad_(True)
    fn = P.get_adversarial_perceptual_loss_fn(model)
    got = fn(pd, xd, tgt.to(dev), target_div=2)
    (got * w.to(dev)).sum().backward()
    got_rep = fn(pd, x.to(dev), tgt2.to(dev))
    assert torch.equal(got.detach(), got_rep)
    feats = model(pd, x.to(dev))
    if dtype == torch.float32:
        for k in P.PERCEPTUAL_LAYERS:
            assert_close(feats[k].reshape(feats_ref[k].shape), feats_ref[k], what=k)
        assert_close(got, ref, rtol=1e-3, atol=1e-5, what="per-sample loss")
        assert_close_scaled(xd.grad, xr.grad, rel=1e-3, what="d loss / d reconstruction")
        assert_close(P.get_perceptual_loss(model, pd, x.to(dev), tgt2.to(dev)), OP.perceptual_loss(p, x, tgt2), rtol=1e-3, atol=1e-5)
    else:
        from test_gpu_parity_r2 import check_bf16
        xe = x.clone().requires_grad_(True)
        emu = OP.adversarial_perceptual_loss(p, xe, tgt2, dtype=torch.bfloat16)
        (emu * w).sum().backward()
        feats_emu = OP.vgg_features(p, x.reshape(16, 32, 32, 3), dtype=torch.bfloat16)
        report = []
        for k in P.PERCEPTUAL_LAYERS:
            check_bf16(k, feats[k].reshape(feats_ref[k].shape), feats_emu[k], feats_ref[k], report)
        check_bf16("loss", got, emu, ref, report)
        check_bf16("dx", xd.grad, xe.grad, xr.grad, report)
        print(report)


@pytest.mark.gpu
def test_rl_loss_with_perceptual_term_vs_oracle(dev):
    """loss_fn with gamma3 * perceptual switched on (rl_nonadversarial.py:125,147): total, the 'perceptual_loss' aux entry and the
    parameter gradients (the term back-propagates through the decoder)."""
    import video_vae_amd as V
    from video_vae_amd import loss as L, perceptual as P, rl_model
    from oracle import loss as OLoss, model as OM
    from test_gpu_parity_r2 import TINY, _load, _step_noise
    cfg = OM.VAEConfig(**TINY)
    p0 = OM.init_video_vae(cfg, seed=3, zero_final=False)
    b, t = 2, 8
    video = torch.rand((b, t, 32, 32, 3), generator=torch.Generator().manual_seed(0))
    mask = torch.ones(b, t); mask[1, 6:] = 0
    emask = OLoss.expand_mask(mask.bool(), cfg.hw)
    noise = _step_noise(cfg, b, t, 0, "rl")
    vp = _params(7)
    hp = dict(OLoss.HPARAMS); hp["gamma3"] = 0.5
    # oracle
    pr = {k: v.clone().requires_grad_(True) for k, v in p0.items()}
    outs = OM.video_vae_rl(pr, cfg, video, emask, noise)
    perc = OP.adversarial_perceptual_loss(vp, outs[0], video.repeat_interleave(2, dim=0))
    ref, ref_aux = OLoss.loss_fn_rl(outs, video, mask, hp, perceptual=perc)
    ref.backward()
    assert float(ref_aux["perceptual_loss"]) > 0
    # product
    m = _load(rl_model.VideoVAE(rngs=V.Rngs(2), dtype=torch.float32, **TINY), p0, dev)
    vgg = P.VGG16Features(dtype=torch.float32)
    fn = P.get_adversarial_perceptual_loss_fn(vgg)
    rngs = V.Rngs(3)
    for k, v in noise.items():
        rngs.inject(k, v)
    hpg = dict(L.HPARAMS); hpg["gamma3"] = 0.5
    loss, aux = L.loss_fn(m, video.to(dev), L.expand_mask(mask.to(dev), cfg.hw), mask.to(dev), rngs, hpg, fn,
                          {k: v.to(dev) for k, v in vp.items()})
    loss.backward()
    assert_close(aux["perceptual_loss"], ref_aux["perceptual_loss"], rtol=1e-3, atol=1e-6, what="perceptual_loss")
    assert_close(loss, ref, rtol=1e-3, atol=1e-6, what="loss")
    from util import grad_floor
    ref_grads = {k: v.grad for k, v in pr.items()}
    for k, prm in m.named_parameters():
        assert_close_scaled(prm.grad, ref_grads[k], rel=2e-3, what=k, floor=grad_floor(k, ref_grads))


@pytest.mark.gpu
def test_train_driver_with_perceptual_term(dev, tmp_path):
    """python -m video_vae_amd.train --vgg random: the driver wires load_vgg / get_adversarial_perceptual_loss_fn into train_step as
    rl_nonadversarial.py:272-274,332 does; the logged perceptual term is positive and the loss finite."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), PYTHONUNBUFFERED="1")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, "-m", "video_vae_amd.train", "--small", "--steps", "3", "--size", "32", "--max_frames", "8", "--vgg", "random"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("Epoch 0, Step")]
    assert lines, out.stdout[-2000:]
    m = re.search(r"Loss = ([-0-9.e+naif]+).*perceptual_loss = ([-0-9.e+naif]+)", lines[-1])
    assert m, lines[-1]
    assert float(m.group(1)) == float(m.group(1)) and float(m.group(2)) > 0
