"""Which parts of the dense weight-gradient GEMM's main loop (gemm_tn256.hip) cost what?  Builds with -DTN_ABL=bits -DTN_STAMPS (tools/r04_tn_abl.sh):
DMA / fragment reads / MFMAs switched off, s_memtime / s_memrealtime stamps around the main loop -> clock and cycles per 32-token stage.
    VVAE_AB_LIB=<variant .so> python tools/tn_ablation.py <label>"""
import ctypes
import os
import sys
import time
sys.path.insert(0, ".")
import torch
import video_vae_amd._lib as _L
_L.LIB_PATH = os.environ["VVAE_AB_LIB"]
from video_vae_amd import ops

dev = "cuda"
torch.manual_seed(0)
K, M, N = 16384, 768, 1536
a = torch.randn(K, M, device=dev, dtype=torch.bfloat16)
b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
# the grouped form the train step uses: 28 such products = 504 whole-K tiles, one per workgroup (two rounds of 256)
n = 28
A, B_ = [a] * n, [b] * n
outs = [torch.empty(M, N, device=dev) for _ in range(n)]
dbs = [torch.empty(N, device=dev) for _ in range(n)]
L = ctypes.CDLL(_L.LIB_PATH)
VP, IA = ctypes.c_void_p * n, ctypes.c_int * n


def run():
    rc = _L.lib().vvae_gemm_tn_grouped_bf16(VP(*[x.data_ptr() for x in A]), IA(*[M] * n), VP(*[x.data_ptr() for x in B_]), IA(*[N] * n),
                                            VP(*[x.data_ptr() for x in outs]), VP(*[x.data_ptr() for x in dbs]), IA(*[M] * n), IA(*[N] * n), n, K,
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc


t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(5):
        run()
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 5 * 1e3
buf = (ctypes.c_ulonglong * 1024)()
assert L.vvae_gemm_tn_stamps(buf) == 0
s = torch.tensor(list(buf), dtype=torch.float64).view(256, 4)
cyc, wall = s[:, 2] - s[:, 0], (s[:, 3] - s[:, 1]) * 10.0
print(f"{sys.argv[1]:12s}: launch {us:7.1f} us ({2.0 * M * N * K * n / us / 1e6:5.0f} TF); main loop of a tile {wall.median().item() / 1e3:6.1f} us, clock {(cyc / wall).median().item():.2f} GHz, "
      f"{cyc.median().item() / (K // 32):6.0f} cycles per 32-token stage (1152 = its two MFMA phases of 36 at 16 cycles)", flush=True)
