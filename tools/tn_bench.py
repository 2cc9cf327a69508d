import torch, sys, os
sys.path.insert(0, ".")
from video_vae_amd import ops
from video_vae_amd._lib import lib
dev = "cuda"
def tmg(f, n=20):
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
K = 16384
torch.manual_seed(0)
for M, N in [(768, 1536), (512, 768), (1536, 768), (768, 768), (768, 512), (256, 256)]:
    a = torch.randn(K, M, device=dev, dtype=torch.bfloat16)
    b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
    want = a.float().t() @ b.float(); wdb = b.float().sum(0)
    res = {}
    for big in (1, 0):
        lib().vvae_gemm_tn_use_big_tiles(big)
        c, db = ops.gemm_tn(a, b, True)
        err = ((c - want).abs().max() / want.abs().max()).item(); errb = ((db - wdb).abs().max() / wdb.abs().max()).item()
        t = tmg(lambda: ops.gemm_tn(a, b, True))
        res[big] = (err, errb, t)
    lib().vvae_gemm_tn_use_big_tiles(1)
    fl = 2.0 * K * M * N
    print(f"M{M} N{N}: big err {res[1][0]:.2e}/{res[1][1]:.2e} {res[1][2]:.1f}us {fl/res[1][2]/1e6:.0f}TF | 128-tile err {res[0][0]:.2e} {res[0][2]:.1f}us {fl/res[0][2]/1e6:.0f}TF", flush=True)
