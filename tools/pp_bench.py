"""gemm_pp.hip against gemm_nt.hip and the library on the trunk's product shapes: bitwise agreement of all four epilogues, then GPU time per
call from replayed hipGraphs of 20 back-to-back calls.   python tools/pp_bench.py"""
import sys
sys.path.insert(0, ".")
import torch
from video_vae_amd import ops
from video_vae_amd._lib import lib

dev = "cuda"


def tmg(f, n=20):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            f()
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                f()
        g.replay(); st.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5):
            g.replay()
        e1.record(st); st.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3


M = 16384
torch.manual_seed(0)
for N, K in [(1536, 768), (768, 1536), (768, 512), (512, 768), (768, 768), (1536, 128)]:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    b = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    bad = []
    for spread in (1,):
        for epi, kw in ((0, dict(bias=bias)), (0, dict()), (1, dict(bias=bias, res=res)), (2, dict(bias=bias)), (3, dict(res=res))):
            want = ops.gemm_nt(a, b, epi=epi, form="nt", **kw)
            got = ops.gemm_nt(a, b, epi=epi, form="pp", **kw)
            torch.cuda.synchronize()
            for w, g_ in zip(want if isinstance(want, tuple) else (want,), got if isinstance(got, tuple) else (got,)):
                if not torch.equal(w, g_):
                    nb = int((w != g_).sum())
                    bad.append((spread, epi, bool(kw.get("bias") is not None), nb, float((w.float() - g_.float()).abs().max())))
    lib().vvae_gemm_pp_ablate(0)
    ref = (a.float() @ b.float().t() + bias)
    err = (ops.gemm_nt(a, b, bias, form="pp").float() - ref).abs().max().item()
    print(f"N{N} K{K}: pp vs fp32 max err {err:.3e}; bitwise mismatches vs gemm_nt: {bad if bad else 'none'}", flush=True)
    fl = 2.0 * M * N * K
    bb = bias.to(torch.bfloat16); bt = b.t()
    t_lib = tmg(lambda: torch.addmm(bb, a, bt))
    row = [f"library addmm {t_lib:6.1f} us ({fl / t_lib / 1e6:5.0f} TF)"]
    for form in ("nt", "pp"):
        ts = [tmg(lambda: ops.gemm_nt(a, b, bias, form=form)), tmg(lambda: ops.gemm_nt(a, b, bias, res, ops.EPI_RES, form=form)),
              tmg(lambda: ops.gemm_nt(a, b, bias, None, ops.EPI_SILU, form=form)), tmg(lambda: ops.gemm_nt(a, b, None, res, ops.EPI_MUL_DSILU, form=form))]
        row.append(f"{form}: plain {ts[0]:6.1f} ({fl / ts[0] / 1e6:5.0f} TF) +res {ts[1]:6.1f} silu-pair {ts[2]:6.1f} *dsilu {ts[3]:6.1f}")
    reps = []
    for _ in range(2):                                             # last epilogue of a launch: unit by unit (0) / through the idle rings (1), alternating
        for fr in (0, 1):
            lib().vvae_gemm_pp_final_ring(fr)
            reps.append((fr, tmg(lambda: ops.gemm_nt(a, b, bias, form="pp")), tmg(lambda: ops.gemm_nt(a, b, None, res, ops.EPI_MUL_DSILU, form="pp"))))
    lib().vvae_gemm_pp_final_ring(1)
    row.append("pp last epilogue (rings?, plain, *dsilu): " + " ".join(f"({fr}, {p0:.1f}, {p3:.1f})" for fr, p0, p3 in reps))
    print("   " + " | ".join(row), flush=True)
