"""The 3x7x7 patch mixer (12 real of 16 channels, reference train/unet.py:98-104) forward / input gradient through the C ABI, weights prepacked.
    python tools/mixer_bench.py            (VVAE_AB_LIB=<other build of the library> for an A/B)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch

import video_vae_amd._lib as _L
if os.environ.get("VVAE_AB_LIB"):
    _L.LIB_PATH = os.environ["VVAE_AB_LIB"]
from video_vae_amd import ops
from conv_bench_util import tmg

dev = "cuda"
N, T, H = 4, 16, 256
torch.manual_seed(0)
x = torch.randn(N, T, H, H, 16, device=dev, dtype=torch.bfloat16); x[..., 12:] = 0
dy = torch.randn(N, T, H, H, 16, device=dev, dtype=torch.bfloat16); dy[..., 12:] = 0
w = torch.zeros(3, 7, 7, 16, 16, device=dev); w[..., :12, :12] = torch.randn(3, 7, 7, 12, 12, device=dev) * 0.05
b = torch.zeros(16, device=dev); b[:12] = torch.randn(12, device=dev)
(pk,) = ops.conv3d_prepack([w], [(12, 12)])
ops.force_generic_conv(True)
y0 = ops.conv3d_fwd_raw(x, w, b).float(); dx0 = ops.conv3d_dgrad_raw(dy, w).float()
ops.force_generic_conv(False)
vox = N * T * H * H
for which in ("fwd", "dgrad"):
    if which == "fwd":
        f = lambda: ops.conv3d_fwd_raw(x, w, b, packed=pk.fwd, k_real=12); ref = y0
    else:
        f = lambda: ops.conv3d_dgrad_raw(dy, w, packed=pk.dgrad, k_real=12); ref = dx0
    out = f().float()
    err = ((out - ref).abs().max() / ref.abs().max()).item()
    t = min(tmg(f) for _ in range(3))
    print(f"mixer {which:5s} 12->12 k7 @{H}: {t:6.1f} us  {vox * 24 * 2 / t / 1e3:5.0f} GB/s (true channels)  {2.0 * vox * 147 * 144 / t / 1e6:5.0f} TF/s (true)  err {err:.1e}", flush=True)
