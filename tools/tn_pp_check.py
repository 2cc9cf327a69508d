"""After the barrier fences: the dense weight-gradient kernel (gemm_tn256) and the NT forms, correctness against fp32 and time per call."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import torch
from video_vae_amd import ops
from pp_bench_util import tmg
dev = "cuda"
torch.manual_seed(0)
K = 16384
for M, N in [(768, 1536), (1536, 768), (512, 768), (768, 768)]:
    a = torch.randn(K, M, device=dev, dtype=torch.bfloat16)
    b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
    assert ops.gemm_tn_supported(a, b)
    dw, db = ops.gemm_tn(a, b, True)
    ref = a.float().t() @ b.float()
    e = float((dw - ref).abs().max() / ref.abs().max())
    eb = float((db - b.float().sum(0)).abs().max() / b.float().sum(0).abs().max())
    t = tmg(lambda: ops.gemm_tn(a, b, True), n=10)
    print(f"gemm_tn {M}x{N} K{K}: rel err {e:.2e} (bias {eb:.2e})  {t:7.1f} us  {2.0 * M * N * K / t / 1e6:6.0f} TF", flush=True)
