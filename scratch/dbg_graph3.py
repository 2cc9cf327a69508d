import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import video_vae_amd as V
from video_vae_amd import optim, loss as L
from video_vae_amd.graph import GraphedTrainStep
TINY = dict(height=32, width=32, channels=3, patch_size=8, encoder_depth=1, decoder_depth=1, mlp_dim=64, num_heads=4,
            qkv_features=32, max_temporal_len=8, spatial_compression_rate=4, unembedding_upsample_rate=4)
dev = torch.device("cuda:0")
ma = V.VideoVAE(rngs=V.Rngs(2), dtype=torch.bfloat16, **TINY).to(dev)
mb = V.VideoVAE(rngs=V.Rngs(2), dtype=torch.bfloat16, **TINY).to(dev)
with torch.no_grad():
    for m in (ma, mb):
        m.decoder.unet.final_conv.kernel.copy_(torch.randn(m.decoder.unet.final_conv.kernel.shape, generator=torch.Generator().manual_seed(5)).to(dev) * 0.2)
oa, ob = optim.Optimizer(ma, 1e-3), optim.Optimizer(mb, 1e-3)
video = torch.rand((2, 8, 32, 32, 3), device=dev).to(torch.bfloat16)
mask = torch.ones(2, 8, device=dev); mask[1, 5:] = 0
ra, rb = V.Rngs(3), V.Rngs(3)
gstep = GraphedTrainStep(ma, oa, video, mask, L.HPARAMS, 16, ra, warmup=1)
with torch.no_grad():
    oa.p.copy_(ob.p); oa.m.copy_(ob.m); oa.v.copy_(ob.v); oa.refresh_shadow()
oa.count = ob.count
fixed = {name: (torch.rand_like(buf) if kind == "uniform" else torch.randn_like(buf)) for name, (kind, buf) in gstep.noise.items()}
gstep._refill = lambda: [buf.copy_(fixed[name]) for name, (kind, buf) in gstep.noise.items()]
for name, t in fixed.items():
    rb.inject(name, t)
for it in range(3):
    lg, _ = gstep()
    le = L.train_step(mb, ob, video, mask, L.HPARAMS, 16, rb)[0]
    print("iter", it, "loss", float(lg), float(le))
    for n, p, off in zip(oa.names, oa.params, oa.offsets):
        a = oa.g[off:off + p.numel()]; b = ob.g[off:off + p.numel()]
        err = float((a - b).abs().max()); sc = float(b.abs().max())
        if not (err <= 2e-3 * sc + 1e-6):
            print(f"   {n:55s} err {err:.3e} scale {sc:.3e}")
