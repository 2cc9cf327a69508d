"""In-kernel clock of the GEMM main loop (MI355X_MICROARCH.md, DVFS give-back item 6): the -DPP_ABLATION build stamps s_memtime (shader
clock) and s_memrealtime (100 MHz) around the main loop of wave 0 of every workgroup; after >= 2 s of back-to-back launches on random data
the median over workgroups of d memtime / d memrealtime x 100 MHz is the clock the chip holds, and d memtime / k-steps the cycles per k-step.
    python tools/pp_clock.py"""
import ctypes
import os
import sys
import time
sys.path.insert(0, ".")
import torch
import video_vae_amd._lib as _L
_L.LIB_PATH = os.environ.get("VVAE_AB_LIB", "video_vae_amd/csrc/build/libvvae_hip_ppabl.so")
from video_vae_amd import ops
from video_vae_amd._lib import lib

dev = "cuda"
M = 16384
torch.manual_seed(0)
L = ctypes.CDLL(_L.LIB_PATH)
names = {0: "all", 1: "no DMA", 2: "no reads", 4: "no MFMA", 3: "MFMA only", 6: "DMA only"}
for N, K in [(768, 1536), (1536, 768)]:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    b = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    for bits, nm in names.items():
        lib().vvae_gemm_pp_ablate(bits)
        t0 = time.time()
        while time.time() - t0 < (2.5 if bits == 0 else 1.0):
            for _ in range(50):
                ops.gemm_nt(a, b, bias, form="pp")
            torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 1024)()
        assert L.vvae_gemm_pp_stamps(buf) == 0
        s = torch.tensor(list(buf), dtype=torch.float64).view(256, 4)
        cyc, wall = s[:, 2] - s[:, 0], (s[:, 3] - s[:, 1]) * 10.0          # ns
        ghz = (cyc / wall).median().item()
        ksteps = (K // 64) * (2 if N == 1536 else 1)
        print(f"N{N} K{K} {nm:10s}: clock {ghz:.2f} GHz, main loop {wall.median().item() / 1e3:6.1f} us = {cyc.median().item() / ksteps:6.0f} cycles per k-step "
              f"({ksteps} k-steps; 1536 = the two MFMA phases at 16 cycles per MFMA)", flush=True)
    lib().vvae_gemm_pp_ablate(0)
