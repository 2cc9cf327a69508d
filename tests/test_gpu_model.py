"""GPU parity of the assembled path: UNet, VideoVAE (both flavours), losses, optimiser step -- HIP vs CPU oracle."""
import copy

import zlib
import pytest
import torch

from oracle import unet as OU
from oracle import model as OM
from oracle import loss as OLoss
from oracle import optim as OOpt
from util import assert_close, assert_close_scaled, grad_floor, rnd

pytestmark = pytest.mark.gpu
GRAD_REL = 1e-3          # fp32 gradients: within 1e-3 of the tensor's scale, element-wise (the generic path is exact-fp32 MFMA: the
                         # same bar as the forward's rtol)

TINY = dict(height=32, width=32, channels=3, patch_size=8, encoder_depth=1, decoder_depth=1, mlp_dim=64, num_heads=4,
            qkv_features=32, max_temporal_len=8, spatial_compression_rate=4, unembedding_upsample_rate=4)


def _load(module, params, dev):
    sd = module.state_dict()
    assert set(sd) == set(params), (set(sd) ^ set(params))
    with torch.no_grad():
        for k, v in params.items():
            sd[k].copy_(v)
    return module.to(dev)


@pytest.mark.parametrize("shape,levels,base", [((1, 8, 64, 64, 12), 3, 16), ((2, 3, 16, 24, 5), 2, 8)])
def test_unet_fwd_bwd_fp32(dev, shape, levels, base):
    """Config C1-sized UNet (B=1, 3x8x64x64 features) fp32: output and every parameter gradient vs the oracle."""
    import video_vae_amd as V
    c = shape[-1]
    p = OU.init_unet(c, base, levels, 3, seed=5, zero_final=False)
    for k in p:
        if k.endswith("bias") or k.endswith("scale"):
            p[k] = p[k] + 0.1 * rnd(p[k].shape, zlib.crc32(k.encode()) % 1000)   # not hash(): str hashes are salted per process
    x = rnd(shape, 40, 0.5)
    gy = rnd(shape[:-1] + (3,), 41)
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xo = x.clone().requires_grad_(True)
    yo = OU.unet(po, xo)
    yo.backward(gy)
    m = _load(V.UNet(c, base, levels, 3, V.Rngs(0), dtype=torch.float32), p, dev)
    xg = x.to(dev).requires_grad_(True)
    yg = m(xg)
    yg.backward(gy.to(dev))
    assert_close(yg, yo, what="unet out")
    assert_close_scaled(xg.grad, xo.grad, rel=GRAD_REL, what="dx")
    ref = {k: v.grad for k, v in po.items()}
    for k, prm in m.named_parameters():
        assert_close_scaled(prm.grad, ref[k], rel=GRAD_REL, what=f"d{k}", floor=grad_floor(k, ref))


def test_unet_bf16_close_to_emulated_oracle(dev):
    import video_vae_amd as V
    p = OU.init_unet(12, 16, 2, 3, seed=6, zero_final=False)
    x = rnd((1, 4, 32, 32, 12), 42, 0.5)
    yo = OU.unet(p, x, torch.bfloat16)
    m = _load(V.UNet(12, 16, 2, 3, V.Rngs(0), dtype=torch.bfloat16), p, dev)
    yg = m(x.to(dev))
    assert yg.dtype == torch.bfloat16
    assert_close(yg, yo, rtol=5e-2, atol=5e-2 * float(yo.abs().max()), what="bf16 unet")


def _noise(cfg, b, t, seed):
    g = torch.Generator().manual_seed(seed)
    return {"gumbel_u": torch.rand((b, t, 1), generator=g),
            "reparam_eps": torch.randn((b, t, cfg.hw, cfg.latent_dim), generator=g),
            "bernoulli_u": torch.rand((2 * b, t, 1, 1), generator=g)}


@pytest.mark.parametrize("flavour", ["model", "rl"])
def test_video_vae_loss_and_grads_fp32(dev, flavour):
    """Full tiny VAE (32x32, patch 8, depth 1): forward tuple, loss terms and all gradients vs the oracle."""
    import video_vae_amd as V
    from video_vae_amd import loss as L, rl_model
    cfg = OM.VAEConfig(**TINY)
    p = OM.init_video_vae(cfg, seed=3, zero_final=False)
    b, t = 2, 8
    video = torch.rand((b, t, 32, 32, 3), generator=torch.Generator().manual_seed(0))
    mask = torch.ones(b, t); mask[1, 6:] = 0
    noise = _noise(cfg, b, t, 7)
    emask = OLoss.expand_mask(mask.bool(), cfg.hw)
    if flavour == "rl":
        # the Bernoulli gate is a hard threshold u < p: keep every injected u at least 0.02 away from the oracle's p so
        # that last-bit differences between the two paths cannot flip a frame (that would be a discrete, O(1) change)
        with torch.no_grad():
            sel = OM.video_vae_rl(p, cfg, video, emask, noise)[2]
        u = noise["bernoulli_u"]
        close = (u - sel).abs() < 0.02
        u = torch.where(close, torch.where(sel > 0.5, sel - 0.05, sel + 0.05), u)
        # ... and make the two members of every pair differ (keep frame 0 in one, drop it in the other): with identical
        # masks the pair's losses are equal up to summation order and the reference's (loss - mean) / (std + 1e-6)
        # turns that rounding noise into an O(1) "disadvantage" -- ill-conditioned by construction, not a parity question
        u[0::2, 0] = 0.0
        u[1::2, 0] = 0.9999
        noise["bernoulli_u"] = u
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    if flavour == "model":
        out_o = OM.video_vae(po, cfg, video, emask, noise)
        loss_o, aux_o = OLoss.loss_fn_plain(out_o, video, mask)
        cls = V.VideoVAE
    else:
        out_o = OM.video_vae_rl(po, cfg, video, emask, noise)
        loss_o, aux_o = OLoss.loss_fn_rl(out_o, video, mask)
        cls = rl_model.VideoVAE
    loss_o.backward()
    m = _load(cls(rngs=V.Rngs(2), dtype=torch.float32, **TINY), p, dev)
    rngs = V.Rngs(3)
    for k, v in noise.items():
        rngs.inject(k, v)
    vg, mg = video.to(dev), mask.to(dev)
    em = L.expand_mask(mg.bool(), cfg.hw)
    if flavour == "model":
        loss_g, aux_g = L.loss_fn_plain(m, vg, em, mg, rngs, L.HPARAMS)
    else:
        loss_g, aux_g = L.loss_fn(m, vg, em, mg, rngs, L.HPARAMS)
    loss_g.backward()
    assert_close(aux_g["reconstruction"], aux_o["reconstruction"], what="reconstruction")
    for k in aux_o:
        if k != "reconstruction":
            assert_close(aux_g[k], aux_o[k], rtol=1e-3, atol=1e-5, what=k)
    assert_close(loss_g, loss_o, rtol=1e-3, atol=1e-5, what="loss")
    ref = {n: v.grad for n, v in po.items()}
    bad = []
    for k, prm in m.named_parameters():
        assert prm.grad is not None, k
        assert torch.isfinite(prm.grad).all(), k
        try:                                   # every tensor is checked before the test fails: the message lists all that miss the bar
            assert_close_scaled(prm.grad, ref[k], rel=GRAD_REL, what=f"d{k}", floor=grad_floor(k, ref))
        except AssertionError as e:
            bad.append(str(e))
    assert not bad, "\n".join(bad)


def test_rl_outputs_contract(dev):
    """Reference shape/binary contracts (claude_distributed/test_rl_model.py:132-136)."""
    import video_vae_amd as V
    from video_vae_amd import rl_model, loss as L
    m = rl_model.VideoVAE(rngs=V.Rngs(2), dtype=torch.float32, **TINY).to(dev)
    b, t = 2, 8
    video = torch.rand((b, t, 32, 32, 3), device=dev)
    em = L.expand_mask(torch.ones(b, t, device=dev).bool(), 16)
    recon, comp, sel, sel_mask, lv, mean = m(video, em, V.Rngs(1))
    assert recon.shape == (2 * b, t, 32, 32, 3) and comp.shape == (2 * b, t, 16, 48)
    assert sel.shape == (2 * b, t, 1, 1) and sel_mask.shape == (2 * b, t, 1, 1)
    assert set(sel_mask.unique().tolist()) <= {0.0, 1.0}


def test_optimizer_steps_match_oracle(dev):
    """clip_by_global_norm + adam + warmup-cosine over 3 updates vs the optax restatement."""
    import video_vae_amd as V
    from video_vae_amd import optim
    m = V.UNet(4, 8, 1, 3, V.Rngs(1), dtype=torch.float32)
    p0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(dev)
    kw = dict(init_value=0.0, peak_value=1e-2, warmup_steps=2, decay_steps=10, end_value=1e-3)
    opt = optim.Optimizer(m, optim.warmup_cosine_decay_schedule(**kw), max_norm=1.0)
    adam = OOpt.Adam(p0)
    po = p0
    for step in range(3):
        grads = {k: rnd(v.shape, 100 + step * 31 + i, 0.3 if step else 5.0) for i, (k, v) in enumerate(p0.items())}
        opt.set_grads(grads)
        opt.update()
        po, gn, lr = OOpt.train_update(po, grads, adam, kw)
        assert abs(opt.grad_norm() - float(gn)) <= 1e-4 * float(gn)
        for k, prm in m.named_parameters():
            assert_close(prm, po[k], rtol=1e-4, atol=1e-6, what=f"step{step} {k}")


def test_loss_decreases_on_fixed_batch(dev):
    """Reference integration property (claude_distributed/test_training_loop.py:168-178): 10 steps, lr 1e-3."""
    import video_vae_amd as V
    from video_vae_amd import optim, loss as L, rl_model
    m = rl_model.VideoVAE(rngs=V.Rngs(2), dtype=torch.float32, **TINY).to(dev)
    opt = optim.Optimizer(m, 1e-3)
    video = torch.rand((2, 8, 32, 32, 3), device=dev); mask = torch.ones(2, 8, device=dev)
    rngs = V.Rngs(3)
    losses = []
    for _ in range(10):
        loss, aux = L.train_step(m, opt, video, mask, L.HPARAMS, 16, rngs)
        assert torch.isfinite(loss)
        losses.append(float(aux["MSE"]))
    assert sum(losses[5:]) / 5 < sum(losses[:5]) / 5, losses


def test_checkpoint_roundtrip(dev, tmp_path):
    import video_vae_amd as V
    from video_vae_amd import optim
    m = V.UNet(4, 8, 1, 3, V.Rngs(1), dtype=torch.float32).to(dev)
    opt = optim.Optimizer(m, 1e-2)
    x = torch.randn(1, 2, 8, 8, 4, device=dev)
    for _ in range(2):
        opt.zero_grad(); m(x).square().mean().backward(); opt.update()
    V.save_checkpoint(m, opt, str(tmp_path / "ck"))
    ref = {k: v.clone() for k, v in m.state_dict().items()}
    m2 = V.UNet(4, 8, 1, 3, V.Rngs(9), dtype=torch.float32).to(dev)
    opt2 = optim.Optimizer(m2, 1e-2)
    assert V.load_checkpoint(m2, opt2, str(tmp_path / "ck")) is None
    for k, v in m2.state_dict().items():
        assert torch.equal(v, ref[k]), k
    assert opt2.count == opt.count and torch.equal(opt2.m, opt.m) and torch.equal(opt2.v, opt.v)
    opt.zero_grad(); m(x).square().mean().backward(); opt.update()
    opt2.zero_grad(); m2(x).square().mean().backward(); opt2.update()
    # wgrad accumulates with fp32 atomics: replicas agree to rounding, not bitwise
    assert torch.allclose(opt.p, opt2.p, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("split,depth,graphs", [(False, 1, 1), (True, 1, 2), (True, 4, 4), (True, 2, 3)])
def test_graphed_train_step_matches_eager(dev, split, depth, graphs):
    """hipGraph replay of forward+backward must produce the eager step's loss and gradients (same weights, same noise);
    split = the staged form used under data parallelism: forward + decoder backward, then the encoder's backward in up to three
    segments (cut at the outputs of encoder blocks; fewer when the encoder is shallower)."""
    import video_vae_amd as V
    from video_vae_amd import optim, loss as L
    from video_vae_amd.graph import GraphedTrainStep
    torch.manual_seed(0)
    cfg = dict(TINY, encoder_depth=depth)
    ma = V.VideoVAE(rngs=V.Rngs(2), dtype=torch.bfloat16, **cfg).to(dev)
    mb = V.VideoVAE(rngs=V.Rngs(2), dtype=torch.bfloat16, **cfg).to(dev)
    with torch.no_grad():
        for m in (ma, mb):
            m.decoder.unet.final_conv.kernel.copy_(rnd(m.decoder.unet.final_conv.kernel.shape, 5, 0.2).to(dev))
    oa, ob = optim.Optimizer(ma, 1e-3), optim.Optimizer(mb, 1e-3)
    assert torch.equal(oa.p, ob.p)
    video = torch.rand((2, 8, 32, 32, 3), device=dev).to(torch.bfloat16)
    mask = torch.ones(2, 8, device=dev); mask[1, 5:] = 0
    ra, rb = V.Rngs(3), V.Rngs(3)
    gstep = GraphedTrainStep(ma, oa, video, mask, L.HPARAMS, 16, ra, warmup=1, split=split)     # runs 1 + 1 + 1 updates on model a
    assert 1 + len(gstep.graphs) == graphs
    if split:                                          # every parameter belongs to exactly one stage; stages own runs of buckets
        assert sorted(i for st in gstep.stage_idx for i in st) == list(range(len(oa.params)))
        assert max(gstep.bucket_stage) == graphs - 1
    with torch.no_grad():                                                         # re-align the two replicas
        oa.p.copy_(ob.p); oa.m.copy_(ob.m); oa.v.copy_(ob.v); oa.refresh_shadow()
    oa.count = ob.count
    fixed = {name: (torch.rand_like(buf) if kind == "uniform" else torch.randn_like(buf)) for name, (kind, buf) in gstep.noise.items()}
    gstep._refill = lambda: [buf.copy_(fixed[name]) for name, (kind, buf) in gstep.noise.items()]
    for name, t in fixed.items():
        rb.inject(name, t)
    for _ in range(3):
        loss_g, aux_g = gstep()
        loss_e, aux_e = L.train_step(mb, ob, video, mask, L.HPARAMS, 16, rb)
        assert torch.isfinite(loss_g)
        assert_close(loss_g, loss_e, rtol=1e-3, atol=1e-4, what="loss")
        # wgrad of the small fp32-generic convs uses fp32 atomics: equal to rounding, not bitwise
        assert_close_scaled(oa.g, ob.g, rel=2e-3, what="flat gradient")
        assert torch.allclose(oa.p, ob.p, rtol=1e-4, atol=1e-5)


def _linear_stack():
    import video_vae_amd as V
    from video_vae_amd import layers as LY

    class Stack(torch.nn.Module):
        def __init__(self):
            super().__init__()
            r = V.Rngs(0)
            dims = [768, 1536, 768, 1536, 512, 768, 1536, 768, 1536, 768, 768, 1536, 1536, 768, 768, 1536, 1536, 768]
            self.lins = torch.nn.ModuleList([LY.Linear(a, b, r) for a, b in zip(dims[:-1], dims[1:])])
            self.norms = torch.nn.ModuleList([LY.LayerNorm(b) for b in dims[1:]])

        def forward(self, x):
            for lin, nrm in zip(self.lins, self.norms):
                x = torch.tanh(nrm(lin(x)))
            return x
    return Stack()


@pytest.mark.gpu
def test_deferred_grouped_weight_gradients_match_per_layer_path(dev):
    """A stack of bf16 Linear + LayerNorm layers: backward inside ops.deferred_wgrad (parked products -> grouped whole-K launch,
    parked dgamma/dbeta folds -> grouped fold, both straight into the flat gradient buffer) == plain backward through the hooks."""
    from video_vae_amd import ops, optim

    Stack = _linear_stack
    torch.manual_seed(0)
    m = Stack().to(dev)
    opt = optim.Optimizer(m, 1e-3)
    x = rnd((1024, 768), 7).to(dev, torch.bfloat16)
    gy = rnd((1024, 768), 8).to(dev, torch.bfloat16)

    def grads(deferred):
        opt.zero_grad()
        out = m(x)
        if deferred:
            with ops.deferred_wgrad(opt):
                out.backward(gy)
        else:
            out.backward(gy)
        for b in range(len(opt.buckets)):
            if not opt.landed[b]:
                opt._land(b)
        return opt.g.clone(), set(opt.external)
    g_plain, ext0 = grads(False)
    g_def, ext1 = grads(True)
    assert not ext0 and len(ext1) >= 2 * len(m.lins) + 2 * len(m.norms) - 2, "weights, biases, LayerNorm scales/biases landed externally"
    assert_close_scaled(g_def, g_plain, rel=2e-5, what="flat gradient buffer, deferred vs plain")
    g_def2, _ = grads(True)
    assert torch.equal(g_def, g_def2), "deterministic"


def _ddp_gpu_worker(rank, world, port, out):
    """Two ranks sharing cuda:0 (gloo transport): the bucketed all-reduce is launched from the landing hooks while the
    deferred, grouped weight gradients are still being flushed into the flat buffer."""
    import os
    import torch.distributed as dist
    import video_vae_amd as V
    from video_vae_amd import ops, optim, ddp, loss as L
    from video_vae_amd.graph import GraphedTrainStep
    dist.init_process_group("gloo", rank=rank, world_size=world, init_method=f"tcp://127.0.0.1:{port}")
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        res = {}
        # ---- (a) Linear/LayerNorm stack: hooks + deferred grouped weight gradients + bucketed all-reduce
        torch.manual_seed(0)
        m = _linear_stack().to(dev)
        opt = optim.Optimizer(m, 1e-3, bucket_bytes=8 << 20)
        xs = [rnd((1024, 768), 20 + r).to(dev, torch.bfloat16) for r in range(world)]
        gy = rnd((1024, 768), 8).to(dev, torch.bfloat16)

        def backward(x):
            opt.zero_grad()
            y = m(x)
            with ops.deferred_wgrad(opt):
                y.backward(gy)
            for b in range(len(opt.buckets)):
                if not opt.landed[b]:
                    opt._land(b)
        want = torch.zeros_like(opt.g)
        for x in xs:                                              # sum of the per-shard gradients, no reducer attached
            backward(x)
            want += opt.g
        reducer = ddp.GradReducer(opt)
        reducer.broadcast_parameters(0)
        backward(xs[rank])
        reducer.finish()
        torch.cuda.synchronize()
        res["stack_got"], res["stack_want"], res["nbuckets"] = opt.g.cpu(), want.cpu(), len(opt.buckets)
        res["layout"] = [(n, v.storage_offset(), v.numel()) for n, v in zip(opt.names, opt.gviews)]
        res["buckets"] = list(opt.buckets)
        # ---- (b) the VAE train step: eager (hooks) then hipGraph replay (all-reduce after the replay)
        torch.manual_seed(0)
        vae = V.VideoVAE(rngs=V.Rngs(2), dtype=torch.bfloat16, **dict(TINY, encoder_depth=3)).to(dev)      # 3 encoder blocks: 4 graphs
        vopt = optim.Optimizer(vae, 1e-3, bucket_bytes=64 << 10)
        vred = ddp.GradReducer(vopt)
        vred.broadcast_parameters(0)
        video = torch.rand((2, 8, 32, 32, 3), generator=torch.Generator().manual_seed(5 + rank)).to(dev, torch.bfloat16)
        mask = torch.ones(2, 8, device=dev)
        rngs = V.Rngs(3 + rank)
        # graph first, eager after: the order bench.py uses (capture wants no earlier pass on another stream)
        gstep = GraphedTrainStep(vae, vopt, video, mask, L.HPARAMS, 16, rngs, warmup=1)
        res["ngraphs"] = 1 + len(gstep.graphs)
        res["stage_buckets"] = [sum(1 for b in gstep.bucket_stage if b == st) for st in range(gstep.nstages)]
        for i in range(2):
            loss, _ = gstep()
            res[f"graph_loss{i}"] = float(loss)
        torch.cuda.synchronize()
        res["p_graph"] = vopt.p.cpu()
        res["vlayout"] = [(n, v.storage_offset(), v.numel()) for n, v in zip(vopt.names, vopt.gviews)]
        for i in range(2):
            loss, _ = L.train_step(vae, vopt, video, mask, L.HPARAMS, 16, rngs)
            res[f"eager_loss{i}"] = float(loss)
        torch.cuda.synchronize()
        res["p_eager"] = vopt.p.cpu()
        # ---- (c) multi-rank resume: rank 0 holds a trained state (moments, count), rank 1 a fresh one; after broadcast_state and
        #      one real update (HIP clip+Adam on the all-reduced gradient) the replicas must be bit-identical
        torch.manual_seed(0)
        tiny = V.VideoVAE(rngs=V.Rngs(2), dtype=torch.float32, **TINY).to(dev)
        topt = optim.Optimizer(tiny, optim.warmup_cosine_decay_schedule(0.0, 1e-3, 10, 100, 1e-4), bucket_bytes=64 << 10)
        v32 = torch.rand((2, 8, 32, 32, 3), generator=torch.Generator().manual_seed(9 + rank)).to(dev)
        if rank == 0:
            for _ in range(3):                                     # "the run that was checkpointed": count = 3, non-zero moments
                L.train_step(tiny, topt, v32, mask, L.HPARAMS, 16, V.Rngs(7))
        tred = ddp.GradReducer(topt)
        tred.broadcast_state(0)
        res["resume_count"] = topt.count
        L.train_step(tiny, topt, v32, mask, L.HPARAMS, 16, V.Rngs(11 + rank))
        torch.cuda.synchronize()
        res["resume_p"], res["resume_m"], res["resume_v"], res["resume_count1"] = topt.p.cpu(), topt.m.cpu(), topt.v.cpu(), topt.count
        torch.save(res, os.path.join(out, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_data_parallel_world2_sharing_one_gpu(dev, tmp_path):
    """world_size 2 on the one GPU of the box: all-reduced gradient = sum of the per-shard gradients (deferred grouped weight
    gradients included), replicas bit-identical after eager and graphed steps (claude_distributed/test_distributed.py:75-97,159-163)."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_ddp_gpu_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert r0["nbuckets"] > 1
    if not torch.equal(r0["stack_got"], r1["stack_got"]):
        bad = [(n, int((r0["stack_got"][o:o + k] != r1["stack_got"][o:o + k]).sum()), k, o) for n, o, k in r0["layout"]
               if not torch.equal(r0["stack_got"][o:o + k], r1["stack_got"][o:o + k])]
        raise AssertionError(f"ranks disagree after the all-reduce: {bad[:6]}")
    # The all-reduced gradient is the sum of the two shard gradients to fp32 rounding (round 1 had to open this check to 5 %: with a
    # second process on the GPU, compiler-formed packed fp32 instructions dropped a subtraction in a few LayerNorm rows per ~1000
    # launches -- tools/nondet_probe.py; the library is built without them now and the check is tight again)
    assert_close_scaled(r0["stack_got"], r0["stack_want"], rel=1e-6, what="all-reduced flat gradient vs sum of shard gradients")
    for k in ("p_graph", "p_eager"):
        if not torch.equal(r0[k], r1[k]):
            bad = [(n, int((r0[k][o:o + c] != r1[k][o:o + c]).sum()), c) for n, o, c in r0["vlayout"] if not torch.equal(r0[k][o:o + c], r1[k][o:o + c])]
            raise AssertionError(f"replicas diverged ({k}): {len(bad)} of {len(r0['vlayout'])} parameters, {bad[:10]}")
        assert torch.isfinite(r0[k]).all()
    assert not torch.equal(r0["p_eager"], r0["p_graph"])
    # the captured step was staged: forward + decoder backward, then three encoder segments, each stage owning buckets of its own
    assert r0["ngraphs"] == 4 and len(r0["stage_buckets"]) == 4 and all(n > 0 for n in r0["stage_buckets"]), r0["stage_buckets"]
    # resume: count and moments travelled with the parameters (reference claude_distributed/distributed_train.py:321-341)
    assert r0["resume_count"] == 3 and r1["resume_count"] == 3 and r0["resume_count1"] == 4 and r1["resume_count1"] == 4
    for k in ("resume_p", "resume_m", "resume_v"):
        assert torch.equal(r0[k], r1[k]), k
    assert float(r0["resume_m"].abs().max()) > 0
    for i in range(2):
        assert r0[f"eager_loss{i}"] != r1[f"eager_loss{i}"]          # different shards
        for r in (r0, r1):
            assert r[f"eager_loss{i}"] == r[f"eager_loss{i}"] and r[f"graph_loss{i}"] == r[f"graph_loss{i}"]   # finite


def test_unet_full_size_is_deterministic(dev):
    """BASELINE's full extent through the whole Conv3d UNet (B=4, 16 x 256 x 256 x 12 features -> 3 channels, bf16): forward and
    every parameter gradient are bitwise reproducible run to run (slab-reduced weight gradients, no float atomics on the bf16
    path), finite, and the elided concat leaves no uninitialised channel behind."""
    import video_vae_amd as V
    torch.manual_seed(0)
    net = V.UNet(channels=12, base_features=16, num_levels=3, out_features=3, rngs=V.Rngs(1), dtype=torch.bfloat16).to(dev)
    with torch.no_grad():
        net.final_conv.kernel.copy_(rnd(tuple(net.final_conv.kernel.shape), 5, 0.2).to(dev))
    g = torch.Generator().manual_seed(3)
    x = (torch.randn((4, 16, 256, 256, 12), generator=g) * 0.5).to(dev, torch.bfloat16)
    gy = torch.randn((4, 16, 256, 256, 3), generator=g).to(dev, torch.bfloat16)
    runs = []
    for _ in range(2):
        net.zero_grad()
        xx = x.clone().requires_grad_(True)
        y = net(xx)
        y.backward(gy)
        runs.append([y.detach().clone(), xx.grad.clone()] + [p.grad.clone() for p in net.parameters()])
    for a, b in zip(*runs):
        assert torch.isfinite(a.float()).all()
        assert torch.equal(a, b)
    assert float(runs[0][0].float().abs().max()) > 0 and float(runs[0][1].float().abs().max()) > 0


def test_graphed_train_step_rl_flavour_captures(dev):
    """The rl_model flavour (Bernoulli pairs + the trajectory-probability term, reference train/rl_nonadversarial.py:150-170)
    must capture as a hipGraph too: its loss may not contain host read-backs (torch.prod's backward has one)."""
    import video_vae_amd as V
    from video_vae_amd import optim, loss as L, rl_model
    from video_vae_amd.graph import GraphedTrainStep
    torch.manual_seed(0)
    m = rl_model.VideoVAE(rngs=V.Rngs(2), dtype=torch.bfloat16, **TINY).to(dev)
    opt = optim.Optimizer(m, 1e-3)
    video = torch.rand((2, 8, 32, 32, 3), device=dev).to(torch.bfloat16)
    mask = torch.ones(2, 8, device=dev); mask[1, 6:] = 0
    gstep = GraphedTrainStep(m, opt, video, mask, L.HPARAMS, 16, V.Rngs(3), warmup=1)
    p0 = opt.p.clone()
    for _ in range(2):
        loss, aux = gstep()
        assert torch.isfinite(loss) and torch.isfinite(aux["rl_loss"])
    assert torch.isfinite(opt.p).all() and not torch.equal(opt.p, p0)


def test_sigterm_checkpoints_and_exits(dev, tmp_path):
    """Pre-emption path of the training driver (reference claude_distributed/distributed_train.py:58-67,489-494): SIGTERM flips a
    flag, the loop stops at the next step, rank 0 writes `checkpoint_sigterm_<epoch>` and the process exits 0; the checkpoint loads
    back into a fresh model + optimizer with a non-zero update count."""
    import os
    import signal
    import subprocess
    import sys
    import time
    import video_vae_amd as V
    from video_vae_amd import optim, rl_model
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), PYTHONUNBUFFERED="1")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, "-m", "video_vae_amd.train", "--small", "--steps", "100000", "--size", "32", "--max_frames", "8",
           "--save_dir", str(tmp_path)]
    proc = subprocess.Popen(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        seen, t0 = [], time.time()
        while time.time() - t0 < 300:                              # wait for the first logged step, then pre-empt
            line = proc.stdout.readline()
            if not line:
                break
            seen.append(line)
            if line.startswith("Epoch 0, Step 10"):
                break
        assert any(l.startswith("Epoch 0, Step") for l in seen), "".join(seen)[-2000:]
        proc.send_signal(signal.SIGTERM)
        out, _ = proc.communicate(timeout=120)
    finally:
        if proc.poll() is None:
            proc.kill()
    assert proc.returncode == 0, out[-2000:]
    ck = tmp_path / "checkpoint_sigterm_0"
    assert (ck / "checkpoint.pt").exists(), (os.listdir(tmp_path), out[-1000:])
    cfg = dict(height=32, width=32, channels=3, patch_size=16, encoder_depth=1, decoder_depth=1, mlp_dim=256, num_heads=4,
               qkv_features=128, max_temporal_len=64, spatial_compression_rate=8, unembedding_upsample_rate=4)
    m = rl_model.VideoVAE(rngs=V.Rngs(5), **cfg).to(dev)
    opt = optim.Optimizer(m, 1e-3)
    V.load_checkpoint(m, opt, str(ck))
    assert opt.count >= 10 and float(opt.m.abs().max()) > 0 and torch.isfinite(opt.p).all()


def test_captured_step_holds_no_memset_node_and_the_census_sees_one(dev):
    """Round 3 root cause of the history-dependent replay: a hipGraph MEMSET node replays garbage on ROCm 7.2, and the
    framework's multi-block reduction zeroes its semaphores with one.  GraphedTrainStep counts the node types of what it captured
    (hipGraphGetNodes / hipGraphNodeGetType) and refuses a graph with a memset node; here: (a) the captured VAE step has kernels, no
    memset; (b) the census does see the memset node of a captured framework reduction with a global reduce (the detector is not blind)."""
    import video_vae_amd as V
    from video_vae_amd import optim, loss as L
    from video_vae_amd.graph import GraphedTrainStep, graph_node_census
    m = V.VideoVAE(rngs=V.Rngs(2), dtype=torch.bfloat16, **TINY).to(dev)
    opt = optim.Optimizer(m, 1e-3)
    video = torch.rand((2, 8, 32, 32, 3), device=dev).to(torch.bfloat16)
    mask = torch.ones(2, 8, device=dev)
    gstep = GraphedTrainStep(m, opt, video, mask, L.HPARAMS, 16, V.Rngs(3), warmup=1)
    c = gstep.census[0]
    assert c is not None, "this build hands out no raw graph: the memset guard is blind"
    assert c.get("kernel", 0) > 50 and c.get("memset", 0) == 0, c
    loss, _ = gstep()
    assert torch.isfinite(loss)
    # (b) a reduction over 16 384 rows x 96 columns splits the rows over workgroups: staging buffer + semaphore + hipMemsetAsync
    x = torch.randn(16384, 96, device=dev, dtype=torch.bfloat16)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        x.sum(0, dtype=torch.float32)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g, stream=s):
        y = x.sum(0, dtype=torch.float32)
    c2 = graph_node_census(g)
    assert c2 is not None and c2.get("memset", 0) >= 1, c2
