"""Where a persistent NT-GEMM workgroup spends its time: a diagnostic build of gemm_nt.hip (-DNT_STAMPS: wave 0 stamps s_memrealtime at
the phase boundaries) run on the production shape after a burst of back-to-back launches.
    hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-slp-vectorize -fno-vectorize -shared -DNT_STAMPS video_vae_amd/csrc/gemm_nt.hip \
        -o tools/_probe/libnt_stamps.so
    [PREFETCH=n] python tools/nt_timeline_probe.py [epilogue kind 0..3]"""
import ctypes, os, sys
import torch
dev = "cuda"
M, N, K = 16384, 1536, 768
epi = int(sys.argv[1]) if len(sys.argv) > 1 else 2
a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
b = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * K ** -0.5
bias = torch.randn(N, device=dev)
res = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
c = torch.empty(M, N, device=dev, dtype=torch.bfloat16); c2 = torch.empty_like(c)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
lib = ctypes.CDLL(os.path.abspath("tools/_probe/libnt_stamps.so"))
if os.environ.get("PREFETCH") is not None:
    lib.vvae_gemm_nt_prefetch(int(os.environ["PREFETCH"])); lib.vvae_gemm_nt_prefetch_mask(15)
def run():
    return lib.vvae_gemm_nt_bf16(P(a), K, P(b), K, P(c), N, P(bias), P(res) if epi in (1, 3) else None, N, P(c2) if epi == 2 else None, N, epi, M, N, K,
                                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
for _ in range(200):
    assert run() == 0
torch.cuda.synchronize()
host = (ctypes.c_ulonglong * (256 * 16))()
assert lib.vvae_nt_stamps_copy(host) == 0
import numpy as np
st = np.array(host, dtype=np.int64).reshape(256, 16)[:, :7].astype(np.float64) / 100.0          # us (100 MHz)
t0 = st[:, 0].min()
names = ["start", "k0 landed (tile 1)", "main loop done", "epilogue issued", "k0 landed (tile 2)", "main loop done", "epilogue issued"]
for i, nm in enumerate(names):
    col = st[:, i] - t0
    print(f"{nm:22s} median {np.median(col):6.2f} us   min {col.min():6.2f}  max {col.max():6.2f}   (delta to previous, median {np.median(st[:, i] - st[:, i - 1]) if i else 0:5.2f})")
