"""Upper bound on hiding the epilogue of a tile in mid-launch: the -DPP_ABLATION build with bit 8 skips those epilogues (wrong output), timed against the
full kernel on the two-tiles-per-CU shape.   python tools/pp_mid_epilogue.py"""
import os
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import torch
import video_vae_amd._lib as _L
_L.LIB_PATH = os.environ.get("VVAE_AB_LIB", "video_vae_amd/csrc/build/libvvae_hip_ppabl.so")
from video_vae_amd import ops
from video_vae_amd._lib import lib
from pp_bench_util import tmg
torch.manual_seed(0)
for M, N, K in [(16384, 1536, 768), (32768, 768, 768), (32768, 1536, 768), (16384, 768, 768)]:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    b = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    row = []
    for bits in (0, 8, 0, 8):
        lib().vvae_gemm_pp_ablate(bits)
        row.append(f"{'full' if bits == 0 else 'no mid epilogue'} {tmg(lambda: ops.gemm_nt(a, b, bias, form='pp')):5.1f}")
    lib().vvae_gemm_pp_ablate(0)
    print(f"M{M} N{N} K{K}: " + " | ".join(row), flush=True)
