"""Does the row pitch of the operands matter to gemm_pp (L2 channel spread of a k-tile's 128-byte row pieces)?  Times the plain product with the token
operand / the weight operand at their natural pitch and padded by 64 / 128 / 192 elements.   python tools/pp_pitch.py"""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import torch
from video_vae_amd import ops
from pp_bench_util import tmg

dev = "cuda"
M = 16384
torch.manual_seed(0)
for N, K in [(1536, 768), (768, 1536), (768, 768), (768, 512), (2304, 768)]:
    row = []
    for pa, pb in [(0, 0), (64, 0), (128, 0), (192, 0), (0, 64), (64, 64), (32, 32), (96, 96)]:
        a = torch.randn(M, K + pa, device=dev, dtype=torch.bfloat16)[:, :K]
        b = (torch.randn(N, K + pb, device=dev) / K ** 0.5).to(torch.bfloat16)[:, :K]
        bias = torch.randn(N, device=dev)
        if ops.lib().vvae_gemm_pp_supported(M, N, K, a.stride(0), b.stride(0), N) == 1:
            t = tmg(lambda: ops.gemm_nt(a, b, bias, form="pp"))
            row.append(f"(+{pa},+{pb}) {t:6.1f}")
        else:
            row.append(f"(+{pa},+{pb}) declined")
    print(f"N{N} K{K}: " + "  ".join(row), flush=True)
