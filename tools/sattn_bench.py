import os, torch, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_vae_amd._lib as _L
if os.environ.get("VVAE_AB_LIB"):        # A/B against another build of the library
    _L.LIB_PATH = os.environ["VVAE_AB_LIB"]
from video_vae_amd import ops

from video_vae_amd.layers import RotaryEmbedding
dev = "cuda"
def tmg(f, n=10):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3): f()
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n): f()
        g.replay(); st.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5): g.replay()
        e1.record(st); st.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3
a, s, heads, d = 64, 256, 8, 64
qkv = torch.randn(a, s, 3 * heads * d, device=dev, dtype=torch.bfloat16)
qs = torch.ones(d, device=dev); ks = torch.ones(d, device=dev)
rope = RotaryEmbedding(d, 256); cos, sin = rope.cos_cached.to(dev).contiguous(), rope.sin_cached.to(dev).contiguous()
do = torch.randn(a, s, heads * d, device=dev, dtype=torch.bfloat16)
out, lse2 = ops.spatial_attn_fwd_raw(qkv, qs, ks, cos, sin, heads)
tf = tmg(lambda: ops.spatial_attn_fwd_raw(qkv, qs, ks, cos, sin, heads))
tb = tmg(lambda: ops.spatial_attn_bwd_raw(qkv, out, lse2, do, qs, ks, cos, sin, heads))
print(f"sattn fwd {tf:.1f} us  bwd(+sum_rows) {tb:.1f} us")
