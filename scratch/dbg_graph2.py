import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import video_vae_amd as V
from video_vae_amd import optim, loss as L
from video_vae_amd.graph import GraphedTrainStep
TINY = dict(height=32, width=32, channels=3, patch_size=8, encoder_depth=1, decoder_depth=1, mlp_dim=64, num_heads=4,
            qkv_features=32, max_temporal_len=8, spatial_compression_rate=4, unembedding_upsample_rate=4)
dev = torch.device("cuda:0")
m = V.VideoVAE(rngs=V.Rngs(2), dtype=torch.bfloat16, **TINY).to(dev)
with torch.no_grad():
    m.decoder.unet.final_conv.kernel.normal_(0, 0.2)
o = optim.Optimizer(m, 0.0)     # lr 0: parameters never move
video = torch.rand((2, 8, 32, 32, 3), device=dev).to(torch.bfloat16)
mask = torch.ones(2, 8, device=dev); mask[1, 5:] = 0
r = V.Rngs(3)
g = GraphedTrainStep(m, o, video, mask, L.HPARAMS, 16, r, warmup=1)
fixed = {name: (torch.rand_like(buf) if kind == "uniform" else torch.randn_like(buf)) for name, (kind, buf) in g.noise.items()}
g._refill = lambda: [buf.copy_(fixed[name]) for name, (kind, buf) in g.noise.items()]
g()
gg = o.g.clone()
# eager with the same noise (buffers are still injected)
o.defer_reduce = False
loss, aux = L.train_step(m, o, video, mask, L.HPARAMS, 16, r)
ge = o.g.clone()
print("loss graph", float(g.loss), "eager", float(loss))
for n, p, off in zip(o.names, o.params, o.offsets):
    a = gg[off:off + p.numel()]; b = ge[off:off + p.numel()]
    err = float((a - b).abs().max()); sc = float(b.abs().max())
    if not (err <= 2e-3 * sc + 1e-6):
        print(f"{n:55s} err {err:.3e} scale {sc:.3e} nan {bool(torch.isnan(a).any())}")
