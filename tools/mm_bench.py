import torch, time, sys
sys.path.insert(0, ".")
from video_vae_amd import ops
dev = "cuda"
def tm(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
K = 16384
for M, N in [(768, 1536), (768, 768), (512, 768), (1536, 768), (768, 512), (768, 2304)]:
    x = torch.randn(K, M, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
    w = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * K * M * N
    t_own = tm(lambda: ops.gemm_tn(x, dy, True))
    t_tn = tm(lambda: torch.mm(x.t(), dy))
    t_fwd = tm(lambda: torch.mm(x, w))
    t_dg = tm(lambda: torch.mm(dy, w.t()))
    try:
        t_f32 = tm(lambda: torch.mm(x.t(), dy, out_dtype=torch.float32))
    except Exception as e:
        t_f32 = float("nan")
    print(f"M{M} N{N}: own {t_own:.1f}us {fl/t_own/1e6:.0f}TF | blas TN bf16 {t_tn:.1f}us {fl/t_tn/1e6:.0f}TF | TN f32out {t_f32:.1f} | fwd NN {t_fwd:.1f}us {fl/t_fwd/1e6:.0f}TF | dgrad NT {t_dg:.1f}us {fl/t_dg/1e6:.0f}TF", flush=True)
print("--- rocblas preferred")
try:
    torch.backends.cuda.preferred_blas_library("cublas")
    for M, N in [(768, 1536), (512, 768)]:
        x = torch.randn(K, M, device=dev, dtype=torch.bfloat16); w = torch.randn(M, N, device=dev, dtype=torch.bfloat16); dy = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
        fl = 2.0 * K * M * N
        t_fwd = tm(lambda: torch.mm(x, w)); t_dg = tm(lambda: torch.mm(dy, w.t()))
        print(f"M{M} N{N}: fwd {t_fwd:.1f}us {fl/t_fwd/1e6:.0f}TF dgrad {t_dg:.1f}us {fl/t_dg/1e6:.0f}TF", flush=True)
except Exception as e:
    print("rocblas path failed", e)
