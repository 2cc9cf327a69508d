import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import video_vae_amd as V
from video_vae_amd import optim, loss as L
from video_vae_amd.graph import GraphedTrainStep
TINY = dict(height=32, width=32, channels=3, patch_size=8, encoder_depth=1, decoder_depth=1, mlp_dim=64, num_heads=4,
            qkv_features=32, max_temporal_len=8, spatial_compression_rate=4, unembedding_upsample_rate=4)
dev = torch.device("cuda:0")
for masked in (False, True):
    m = V.VideoVAE(rngs=V.Rngs(2), dtype=torch.bfloat16, **TINY).to(dev)
    o = optim.Optimizer(m, 1e-3)
    video = torch.rand((2, 8, 32, 32, 3), device=dev).to(torch.bfloat16)
    mask = torch.ones(2, 8, device=dev)
    if masked: mask[1, 5:] = 0
    r = V.Rngs(3)
    l0 = float(L.train_step(m, o, video, mask, L.HPARAMS, 16, r)[0])
    print("eager", l0)
    g = GraphedTrainStep(m, o, video, mask, L.HPARAMS, 16, r, warmup=1)
    print("captured loss", float(g.loss), {k: float(v) for k, v in g.aux.items()})
    for i in range(3):
        l, a = g()
        print("replay", i, float(l), {k: float(v) for k, v in a.items()})
