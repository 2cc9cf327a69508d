import torch, sys
sys.path.insert(0, ".")
from video_vae_amd import ops
import torch.nn.functional as F
dev = "cuda"
def tm(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rows, c in [(16384, 768), (16384, 1536), (16384, 512)]:
    x = torch.randn(rows, c, device=dev, dtype=torch.bfloat16)
    g = torch.randn(c, device=dev); b = torch.randn(c, device=dev)
    dy = torch.randn(rows, c, device=dev, dtype=torch.bfloat16)
    t_f = tm(lambda: ops.layer_norm(x, g, b))
    t_t = tm(lambda: F.layer_norm(x, (c,), g.bfloat16(), b.bfloat16(), 1e-6))
    xg = x.clone().requires_grad_(True); gg = g.clone().requires_grad_(True); bg = b.clone().requires_grad_(True)
    y = ops.layer_norm(xg, gg, bg)
    t_b = tm(lambda: torch.autograd.grad(y, (xg, gg, bg), dy, retain_graph=True))
    t_add = tm(lambda: x + dy)
    print(f"rows {rows} C {c}: own fwd {t_f:.1f} us ({2*rows*c*2/t_f/1e6:.2f} TB/s) torch fwd {t_t:.1f} us | own bwd(+sum) {t_b:.1f} us | add {t_add:.1f} us", flush=True)
