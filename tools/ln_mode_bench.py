"""LayerNorm forward: early-stage (__syncthreads before the rows) vs late-stage (raw s_barrier behind the first rows' loads)."""
import sys
import torch
sys.path.insert(0, ".")
from video_vae_amd import ops
from video_vae_amd._lib import lib
dev = "cuda"


def tm(f, n=200):
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


x = torch.randn(16384, 768, device=dev, dtype=torch.bfloat16)
o = torch.randn(16384, 768, device=dev, dtype=torch.bfloat16)
g = torch.randn(768, device=dev); b = torch.randn(768, device=dev)
for rnd in range(3):
    for late in (0, 1):
        lib().vvae_layernorm_fwd_mode(late)
        t1 = tm(lambda: ops.layer_norm(x, g, b))
        t2 = tm(lambda: ops.add_layer_norm_fork(x, o, g, b))
        print(f"round {rnd} late={late}: layernorm_fwd {t1:.2f} us, add_layernorm_fwd {t2:.2f} us", flush=True)
lib().vvae_layernorm_fwd_mode(0)
