import torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from video_vae_amd import ops
dev = "cuda"
M = 16384
N, K = int(sys.argv[1]), int(sys.argv[2])
which = sys.argv[3] if len(sys.argv) > 3 else "own"
torch.manual_seed(0)
a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
b = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
bias = torch.randn(N, device=dev)
bb = bias.bfloat16(); bt = b.t()
for _ in range(10):
    if which == "own":
        ops.gemm_nt(a, b, bias)
    elif which == "tn":
        ops.gemm_tn(a, torch.randn(M, N, device=dev, dtype=torch.bfloat16), True)
    else:
        torch.addmm(bb, a, bt)
torch.cuda.synchronize()
