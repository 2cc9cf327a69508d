"""GPU time of gemm_pp's four epilogues on the trunk's product shapes for ONE build of the library (VVAE_AB_LIB=path selects it): the A/B leg of
tools/r04_pp_variants.sh, which builds gemm_pp.hip with different -DPP_LATE / -DPP_CDMA.   python tools/pp_variants.py label"""
import os
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import torch
import video_vae_amd._lib as _L
if os.environ.get("VVAE_AB_LIB"):
    _L.LIB_PATH = os.environ["VVAE_AB_LIB"]
from video_vae_amd import ops
from pp_bench_util import tmg

label = sys.argv[1] if len(sys.argv) > 1 else "default"
M = 16384
torch.manual_seed(0)
tot = 0.0
cells = []
for N, K, cnt in [(1536, 768, (41, 0, 41, 41)), (768, 1536, (41, 41, 0, 0)), (768, 768, (85, 41, 0, 0)), (512, 768, (41, 0, 0, 0))]:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    b = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
    want = a.float() @ b.float().t() + bias
    got = ops.gemm_nt(a, b, bias, form="pp").float()
    err = ((got - want).abs().max() / want.abs().max()).item()
    ts = [tmg(lambda: ops.gemm_nt(a, b, bias, form="pp")), tmg(lambda: ops.gemm_nt(a, b, bias, res, ops.EPI_RES, form="pp")),
          tmg(lambda: ops.gemm_nt(a, b, bias, None, ops.EPI_SILU, form="pp")), tmg(lambda: ops.gemm_nt(a, b, None, res, ops.EPI_MUL_DSILU, form="pp"))]
    tot += sum(t * c for t, c in zip(ts, cnt))
    cells.append(f"N{N} K{K} e{err:.0e}: " + " ".join(f"{t:5.1f}" for t in ts))
print(f"{label:10s} | " + " | ".join(cells) + f" | weighted by the step's launch counts {tot / 1e3:6.2f} ms", flush=True)
