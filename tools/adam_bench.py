"""Fused clip + Adam update over a flat buffer of the production model's size.   [VVAE_AB_LIB=<other build>] python tools/adam_bench.py"""
import os, sys
sys.path.insert(0, ".")
import torch
import video_vae_amd._lib as _L
if os.environ.get("VVAE_AB_LIB"):
    _L.LIB_PATH = os.environ["VVAE_AB_LIB"]
from video_vae_amd import optim
sys.path.insert(0, "tools")
from conv_bench_util import tmg
n = 170_631_304
m = torch.nn.Linear(1, 1)
m.weight = torch.nn.Parameter(torch.randn(n // 4, 4))
m.bias = None
m = m.cuda()
opt = optim.Optimizer(m, 1e-4)
opt.g.normal_()
t = tmg(lambda: opt.update(), n=5)
bytes_ = n * (4 * 4 + 3 * 4 + 2) + n * 4          # adam: g p m v in, p m v + bf16 out; + the squared-norm pass over g
print(f"update (sqnorm + clip + Adam + shadow) of {n / 1e6:.1f} M parameters: {t:.1f} us = {bytes_ / t / 1e6:.2f} TB/s over {bytes_ / 1e9:.2f} GB", flush=True)
