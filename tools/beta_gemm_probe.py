"""Would folding the residual add into the producing GEMM (D = A B + C + bias, beta = 1) pay?  Times, on the production shapes:
Linear (library, bias epilogue), the same product accumulating into the residual stream in place (beta = 1), LayerNorm alone and
LayerNorm with the pending residual add (what runs today)."""
import sys
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from video_vae_amd import ops
from conv_bench_util import tmg

dev = "cuda"
M = 16384
for K in (512, 1536):
    N = 768
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(K, N, device=dev, dtype=torch.bfloat16) * 0.02
    b = torch.randn(N, device=dev, dtype=torch.bfloat16)
    res = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    with torch.no_grad():
        t_lin = tmg(lambda: torch.addmm(b, x, w, out=out))
        t_acc = tmg(lambda: res.addmm_(x, w))
        t_mm = tmg(lambda: torch.mm(x, w, out=out))
    print(f"K={K}: addmm(bias) {t_lin:6.1f} us   in-place beta=1 {t_acc:6.1f} us   plain mm {t_mm:6.1f} us", flush=True)
sc = torch.ones(768, device=dev); bi = torch.zeros(768, device=dev)
skip = torch.randn(M, 768, device=dev, dtype=torch.bfloat16)
o = torch.randn(M, 768, device=dev, dtype=torch.bfloat16)
with torch.no_grad():
    t_ln = tmg(lambda: ops.layer_norm(skip, sc, bi))
    t_aln = tmg(lambda: ops.add_layer_norm_fork(skip, o, sc, bi))
print(f"LayerNorm {t_ln:6.1f} us   add + LayerNorm (+ sum out) {t_aln:6.1f} us")
