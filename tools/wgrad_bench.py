"""Per-layer Conv3d weight-gradient timing through the C-ABI.  usage: python tools/wgrad_bench.py [cob16 list] [blocks list]"""
import sys
import torch
sys.path.insert(0, ".")
from video_vae_amd import ops
from video_vae_amd._lib import lib
import os, video_vae_amd._lib as _L
if os.environ.get("VVAE_AB_LIB"):        # A/B against another build of the library (tools only)
    _L.LIB_PATH = os.environ["VVAE_AB_LIB"]
sys.path.insert(0, "tools")
from conv_bench_util import tmg

cobs = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,1").split(",")]
blocks = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "768").split(",")]
N, T = 4, 16
shapes = [(16, 16, (3, 7, 7), 256), (16, 16, (3, 3, 3), 256), (32, 16, (3, 3, 3), 256), (32, 32, (3, 3, 3), 128), (16, 32, (3, 3, 3), 128),
          (64, 32, (3, 3, 3), 128), (64, 64, (3, 3, 3), 64), (32, 64, (3, 3, 3), 64), (128, 64, (3, 3, 3), 64), (128, 128, (3, 3, 3), 32)]
torch.manual_seed(0)
for cin, cout, k, H in shapes:
    x = torch.randn(N, T, H, H, cin, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn(N, T, H, H, cout, device="cuda", dtype=torch.bfloat16)
    ks = (*k, cin, cout)
    lib().vvae_conv3d_wgrad_config(0, 0)
    dw0, db0 = ops.conv3d_wgrad_raw(x, dy, ks)
    vox = N * T * H * H
    fl = 2.0 * vox * k[0] * k[1] * k[2] * cin * cout
    line = f"wgrad {cin:3d}->{cout:3d} k{k[1]} @{H}:"
    for c in cobs:
        for nb in blocks:
            lib().vvae_conv3d_wgrad_config(c, nb)
            dw, db = ops.conv3d_wgrad_raw(x, dy, ks)
            err = ((dw - dw0).abs().max() / dw0.abs().max()).item()
            t = tmg(lambda: ops.conv3d_wgrad_raw(x, dy, ks))
            line += f" | c{c}/b{nb} {t:6.1f}us {fl / t / 1e6:4.0f}TF e{err:.0e}"
    print(line, flush=True)
lib().vvae_conv3d_wgrad_config(0, 0)
