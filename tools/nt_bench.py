import torch, sys
sys.path.insert(0, ".")
from video_vae_amd import ops
import torch.nn.functional as F
dev = "cuda"
def tm(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def tmg(f, n=20):
    """GPU time per call from a replayed hipGraph of n calls (no CPU launch overhead)."""
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
torch.manual_seed(0)
for N, K in [(768, 2304), (1536, 768), (768, 512), (768, 1536), (512, 768), (768, 768)]:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    b = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    assert ops.gemm_nt_supported(a, b)
    c = ops.gemm_nt(a, b, bias)
    ref = (a.float() @ b.float().t() + bias)
    err = (c.float() - ref).abs().max().item()
    refb = ref.to(torch.bfloat16)
    mism = (c != refb).float().mean().item()
    c1 = ops.gemm_nt(a, b, bias, res, ops.EPI_RES)
    e1 = (c1.float() - (refb.float() + res.float())).abs().max().item()
    c2, h = ops.gemm_nt(a, b, bias, None, ops.EPI_SILU)
    e2 = (c2.float() - F.silu(h.float())).abs().max().item(); e2h = (h != c).float().mean().item()
    c3 = ops.gemm_nt(a, b, None, res, ops.EPI_MUL_DSILU)
    base = ops.gemm_nt(a, b, None)
    s = torch.sigmoid(res.float()); want3 = base.float() * (s * (1 + res.float() * (1 - s)))
    e3 = (c3.float() - want3).abs().max().item()
    fl = 2.0 * M * N * K
    t_own = tmg(lambda: ops.gemm_nt(a, b, bias))
    t_res = tmg(lambda: ops.gemm_nt(a, b, bias, res, ops.EPI_RES))
    bb = bias.to(torch.bfloat16); bt = b.t()
    t_blas = tmg(lambda: torch.addmm(bb, a, bt))
    t_silu = tmg(lambda: ops.gemm_nt(a, b, bias, None, ops.EPI_SILU))
    t_dsilu = tmg(lambda: ops.gemm_nt(a, b, None, res, ops.EPI_MUL_DSILU))
    hb = torch.addmm(bb, a, bt)
    t_blas_silu = tmg(lambda: F.silu(torch.addmm(bb, a, bt)))
    go = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    t_blas_dsilu = tmg(lambda: torch.ops.aten.silu_backward(torch.mm(a, bt), res))
    print(f"   fused silu {t_silu:.1f}us vs blas+silu {t_blas_silu:.1f}us | fused mul_dsilu {t_dsilu:.1f}us vs blas mm + silu_backward {t_blas_dsilu:.1f}us", flush=True)
    print(f"N{N} K{K}: maxerr {err:.3e} mismatch-vs-rounded-fp32 {mism:.4f} res {e1:.2e} silu {e2:.2e}/{e2h:.4f} dsilu {e3:.2e} | own {t_own:.1f}us {fl/t_own/1e6:.0f}TF  own+res {t_res:.1f}us | blas addmm {t_blas:.1f}us {fl/t_blas/1e6:.0f}TF", flush=True)
