import torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from video_vae_amd import ops
from video_vae_amd.layers import RotaryEmbedding
dev = "cuda"
b, t, hw, heads, d = 4, 16, 256, 8, 64
qkv = torch.randn(b, t, hw, 3 * heads * d, device=dev, dtype=torch.bfloat16, requires_grad=True)
qs = torch.ones(d, device=dev, requires_grad=True); ks = torch.ones(d, device=dev, requires_grad=True)
rope = RotaryEmbedding(d, 64); cos, sin = rope.cos_cached.to(dev).contiguous(), rope.sin_cached.to(dev).contiguous()
go = torch.randn(b, t, hw, heads * d, device=dev, dtype=torch.bfloat16)
for _ in range(6):
    o = ops.temporal_attention_core(qkv, qs, ks, cos, sin, None, 1, heads, 1e-6, inner=hw)
    o.backward(go)
torch.cuda.synchronize()
