"""Where the fused spatial-attention kernels spend their time: attn_spatial.hip built with phases cut out (-DSATTN_PROBE=mask,
bit 0 = no staging, bit 1 = no phase A (fwd: no main loop), bit 2 = no phase B) into tools/_probe/libsattn_<mask>.so; each variant is
timed on the production shape through the C ABI (hipGraph replay).  Timing only: the cut variants compute garbage.
    for m in 0 1 2 4 6; do hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-slp-vectorize -fno-vectorize -shared \
        -DSATTN_PROBE=$m video_vae_amd/csrc/attn_spatial.hip -o tools/_probe/libsattn_$m.so; done
"""
import ctypes, glob, os, sys
import torch
sys.path.insert(0, ".")
from video_vae_amd.layers import RotaryEmbedding

dev = "cuda"
a, s, heads, d = 64, 256, 8, 64
qkv = torch.randn(a * s, 3 * heads * d, device=dev, dtype=torch.bfloat16)
out = torch.empty(a * s, heads * d, device=dev, dtype=torch.bfloat16)
do = torch.randn(a * s, heads * d, device=dev, dtype=torch.bfloat16)
dqkv = torch.empty_like(qkv)
lse = torch.zeros(a * heads, s, device=dev)
part = torch.empty(a * heads, 2, d, device=dev)
qs = torch.ones(d, device=dev); ks = torch.ones(d, device=dev)
rope = RotaryEmbedding(d, 256); cos, sin = rope.cos_cached.to(dev).contiguous(), rope.sin_cached.to(dev).contiguous()
P = lambda t: ctypes.c_void_p(t.data_ptr())


def tmg(f, n=10):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3): f(st.cuda_stream)
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n): f(torch.cuda.current_stream().cuda_stream)
        g.replay(); st.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5): g.replay()
        e1.record(st); st.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3


for path in sorted(glob.glob("tools/_probe/libsattn_*.so")):
    lib = ctypes.CDLL(os.path.abspath(path))
    lib.vvae_spatial_attn_fwd.restype = ctypes.c_int
    lib.vvae_spatial_attn_bwd.restype = ctypes.c_int
    fwd = lambda stream: lib.vvae_spatial_attn_fwd(P(qkv), 3 * heads * d, P(out), heads * d, P(lse), P(qs), P(ks), P(cos), P(sin), a, s, heads, d,
                                                   ctypes.c_float(1e-6), 1, ctypes.c_void_p(stream))
    bwd = lambda stream: lib.vvae_spatial_attn_bwd(P(qkv), 3 * heads * d, P(out), heads * d, P(do), heads * d, P(lse), P(dqkv), 3 * heads * d, P(qs),
                                                   P(ks), P(cos), P(sin), P(part), a, s, heads, d, ctypes.c_float(1e-6), 1, ctypes.c_void_p(stream))
    assert fwd(torch.cuda.current_stream().cuda_stream) == 0 and bwd(torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    print(f"{os.path.basename(path):18s} fwd {tmg(fwd):7.1f} us   bwd {tmg(bwd):7.1f} us", flush=True)
