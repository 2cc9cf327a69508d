import torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from video_vae_amd import ops
from oracle import layers as OL
dev = "cuda"
a, s, heads, d = 64, 256, 8, 64
qkv = torch.randn(a, s, 3 * heads * d, device=dev, dtype=torch.bfloat16)
qs = torch.ones(d, device=dev); ks = torch.ones(d, device=dev)
cos, sin = OL.rope_tables(d, 256); cos, sin = cos.to(dev), sin.to(dev)
do = torch.randn(a, s, heads * d, device=dev, dtype=torch.bfloat16)
for _ in range(6):
    out, lse2 = ops.spatial_attn_fwd_raw(qkv, qs, ks, cos, sin, heads)
    ops.spatial_attn_bwd_raw(qkv, out, lse2, do, qs, ks, cos, sin, heads)
torch.cuda.synchronize()
