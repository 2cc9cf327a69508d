"""Time ops.encoder_head forward / backward at the production shape (B=4, T=16, hw=256, ld=96) with HIP events.

    python tools/head_bench.py [iters]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from video_vae_amd import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda:0")
b, t, hw, ld = 4, 16, 256, 96
g = torch.Generator().manual_seed(0)
bf = torch.bfloat16
mean = (torch.randn(b, t, hw, ld, generator=g) * 0.5).to(dev, bf).requires_grad_(True)
v = (torch.randn(b, t, hw, ld, generator=g) * 1.5).to(dev, bf).requires_grad_(True)
w1 = (torch.randn(ld, 1, generator=g) * ld ** -0.5).to(dev).requires_grad_(True)
b1 = torch.zeros(1, device=dev, requires_grad=True)
w2 = (torch.randn(hw, 1, generator=g) * hw ** -0.5).to(dev).requires_grad_(True)
b2 = torch.zeros(1, device=dev, requires_grad=True)
fill = (torch.randn(1, 1, 1, ld, generator=g) * 0.02).to(dev).requires_grad_(True)
u = torch.rand(b, t, 1, generator=g).to(dev)
eps = torch.randn(b, t, hw, ld, generator=g).to(dev)
mask = torch.ones(b, t, device=dev)
gc = torch.randn(b, t, hw, ld, generator=g).to(dev, bf)
gs = torch.randn(b, t, 1, 1, generator=g).to(dev)
gk = torch.randn(b, t, generator=g).to(dev)
leaves = [mean, v, w1, b1, w2, b2, fill]


def run(n):
    tf = tb = 0.0
    for _ in range(n):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        lv, comp, sel, kl = ops.encoder_head(*leaves, u, eps, mask)
        e[1].record()
        torch.autograd.grad([comp, sel, kl], leaves, [gc, gs, gk])
        e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    return tf / n * 1e3, tb / n * 1e3


run(5)
f, bw = run(iters)
print(f"encoder_head B={b} T={t} hw={hw} ld={ld}: forward {f:.1f} us, backward (kernel + unparked folds) {bw:.1f} us (events around the eager calls)")
