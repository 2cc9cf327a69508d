"""Which parts of the GEMM main loop serialise?  gemm_pp.hip with its DMA, its fragment reads and its MFMAs switched off one by one and in
pairs (timing only: results are garbage), GPU time per call from replayed graphs of 20 back-to-back calls.   python tools/pp_ablation.py"""
import sys
sys.path.insert(0, ".")
import os
import torch
import video_vae_amd._lib as _L
_L.LIB_PATH = os.environ.get("VVAE_AB_LIB", "video_vae_amd/csrc/build/libvvae_hip_ppabl.so")       # built with -DPP_ABLATION (tools/r04c.sh)
from video_vae_amd import ops
from video_vae_amd._lib import lib
from pp_bench_util import tmg

dev = "cuda"
M = 16384
torch.manual_seed(0)
names = {0: "all", 2: "no DMA", 4: "no reads", 8: "no MFMA", 6: "MFMA only", 10: "reads only", 12: "DMA only", 14: "barriers only"}
for N, K in [(768, 1536), (1536, 768), (512, 768)]:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    b = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    row = []
    for bits, nm in names.items():
        lib().vvae_gemm_pp_ablate(bits >> 1)
        t = tmg(lambda: ops.gemm_nt(a, b, bias, form="pp"))
        row.append(f"{nm} {t:5.1f}")
    lib().vvae_gemm_pp_ablate(0)
    print(f"N{N} K{K} ({K // 64} k-steps x {M * N // (256 * (192 if N % 192 == 0 else 128)) // 256 or 1} tiles per CU): " + " | ".join(row), flush=True)
