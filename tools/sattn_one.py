import torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from video_vae_amd import ops
from video_vae_amd.layers import RotaryEmbedding
dev = "cuda"
a, s, heads, d = 64, 256, 8, 64
qkv = torch.randn(a, s, 3 * heads * d, device=dev, dtype=torch.bfloat16)
qs = torch.ones(d, device=dev); ks = torch.ones(d, device=dev)
rope = RotaryEmbedding(d, 256); cos, sin = rope.cos_cached.to(dev).contiguous(), rope.sin_cached.to(dev).contiguous()
do = torch.randn(a, s, heads * d, device=dev, dtype=torch.bfloat16)
for _ in range(6):
    out, lse2 = ops.spatial_attn_fwd_raw(qkv, qs, ks, cos, sin, heads)
    ops.spatial_attn_bwd_raw(qkv, out, lse2, do, qs, ks, cos, sin, heads)
torch.cuda.synchronize()
