"""Where does a persistent NT-GEMM launch spend its time?  Times the production products with three builds of the library:
as shipped, with the epilogue's global stores removed (NT_EXP=1) and with the whole epilogue removed (NT_EXP=2); results of the
probe builds are garbage.   VVAE_AB_LIB=<variant .so> python tools/nt_epilogue_probe.py"""
import os, sys
sys.path.insert(0, ".")
import torch
import video_vae_amd._lib as _L
if os.environ.get("VVAE_AB_LIB"):
    _L.LIB_PATH = os.environ["VVAE_AB_LIB"]
from video_vae_amd import ops
sys.path.insert(0, "tools")
from conv_bench_util import tmg
M = 16384
torch.manual_seed(0)
for N, K in [(1536, 768), (768, 1536), (768, 512), (512, 768)]:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    b = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
    bb, bt = bias.to(torch.bfloat16), b.t()
    t0 = tmg(lambda: ops.gemm_nt(a, b, bias))
    t1 = tmg(lambda: ops.gemm_nt(a, b, bias, res, ops.EPI_RES))
    t2 = tmg(lambda: ops.gemm_nt(a, b, bias, None, ops.EPI_SILU))
    t3 = tmg(lambda: ops.gemm_nt(a, b, None, res, ops.EPI_MUL_DSILU))
    tl = tmg(lambda: torch.addmm(bb, a, bt))
    fl = 2.0 * M * N * K
    print(f"N {N:4d} K {K:4d}: plain {t0:6.1f} us ({fl / t0 / 1e6:4.0f} TF) | +res {t1:6.1f} | silu pair {t2:6.1f} | *dsilu {t3:6.1f} | library addmm {tl:6.1f} ({fl / tl / 1e6:4.0f} TF)", flush=True)
