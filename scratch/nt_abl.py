import torch, sys, os
sys.path.insert(0, ".")
from video_vae_amd import ops
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
M = 16384
for N, K in [(768, 768), (768, 1536), (1536, 768)]:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    b = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    print(f"dbg={os.environ.get('VVAE_NT_DBG','0')} N{N} K{K}: {tmg(lambda: ops.gemm_nt(a, b, None)):.1f} us", flush=True)
