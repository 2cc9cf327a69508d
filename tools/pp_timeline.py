"""Where the time of ONE gemm_pp launch goes (the -DPP_ABLATION build stamps the 100 MHz wall counter per workgroup): dispatch skew, prologue (first
k-tiles in LDS), main loop, final epilogue issue, stores acknowledged -- for waves 0 (first half) and 4 (second half) of every workgroup -- against the
launch's duration in a replayed graph of back-to-back launches.      python tools/pp_timeline.py"""
import ctypes
import os
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import torch
import video_vae_amd._lib as _L
_L.LIB_PATH = os.environ.get("VVAE_AB_LIB", "video_vae_amd/csrc/build/libvvae_hip_ppabl.so")
from video_vae_amd import ops
from pp_bench_util import tmg

M = 16384
torch.manual_seed(0)
L = ctypes.CDLL(_L.LIB_PATH)
for N, K, epi in [(768, 768, 0), (768, 1536, 0), (1536, 768, 0), (1536, 768, 2), (1536, 768, 3), (768, 1536, 1), (512, 768, 0)]:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    b = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
    kw = {0: dict(bias=bias), 1: dict(bias=bias, res=res), 2: dict(bias=bias), 3: dict(res=res)}[epi]
    f = lambda: ops.gemm_nt(a, b, epi=epi, form="pp", **kw)
    t = tmg(f)
    for _ in range(200):                 # hot chip, then one more launch whose stamps are read
        f()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 2048)()
    assert L.vvae_gemm_pp_timeline(buf) == 0
    s = torch.tensor(list(buf), dtype=torch.float64).view(256, 8) * 0.01          # us
    t0 = s[:, 0].min()
    rel = s - t0
    med = lambda c: rel[:, c].median().item()
    print(f"N{N} K{K} epi{epi}: {t:5.1f} us per launch in a graph | workgroup entry: first 0, median {med(0):4.1f}, last {rel[:, 0].max().item():4.1f} | "
          f"loop starts {med(1):5.1f} | waves 0-3: loop ends {med(2):5.1f}, epilogue issued {med(3):5.1f}, stores done {med(4):5.1f} | "
          f"waves 4-7: loop ends {med(5):5.1f}, epilogue issued {med(6):5.1f}, stores done {med(7):5.1f} (last workgroup {rel[:, 7].max().item():5.1f})", flush=True)
