"""Per-layer Conv3d fwd / dgrad timing through the C-ABI, across the rolling-kernel tuning variants.
usage: python tools/conv_bench.py [variants, e.g. 0,1,2,3] [tchunks, e.g. 0,8]"""
import sys
import torch
sys.path.insert(0, ".")
from video_vae_amd import ops
from video_vae_amd._lib import lib
import os, video_vae_amd._lib as _L
if os.environ.get("VVAE_AB_LIB"):        # A/B against another build of the library (tools only)
    _L.LIB_PATH = os.environ["VVAE_AB_LIB"]

dev = "cuda"


sys.path.insert(0, "tools")
from conv_bench_util import tmg

variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,1,2,3").split(",")]
tchunks = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0").split(",")]
N, T = 4, 16
# (cin, cout, (kt,kh,kw), H): fwd runs CK=cin -> CO=cout, dgrad runs CK=cout -> CO=cin
shapes = [(16, 16, (3, 7, 7), 256), (16, 16, (3, 3, 3), 256), (32, 16, (3, 3, 3), 256), (32, 32, (3, 3, 3), 128),
          (16, 32, (3, 3, 3), 128), (64, 32, (3, 3, 3), 128), (32, 64, (3, 3, 3), 64), (64, 64, (3, 3, 3), 64), (128, 64, (3, 3, 3), 64),
          (64, 128, (3, 3, 3), 32), (128, 128, (3, 3, 3), 32)]
torch.manual_seed(0)
for cin, cout, k, H in shapes:
    x = torch.randn(N, T, H, H, cin, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(N, T, H, H, cout, device=dev, dtype=torch.bfloat16)
    w = torch.randn(*k, cin, cout, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    lib().vvae_conv3d_roll_config(0, 0)
    y0 = ops.conv3d_fwd_raw(x, w, b).float()
    dx0 = ops.conv3d_dgrad_raw(dy, w).float()
    vox = N * T * H * H
    fl = 2.0 * vox * k[0] * k[1] * k[2] * cin * cout
    by = vox * (cin + cout) * 2
    for which in ("fwd", "dgrad"):
        ck, co = (cin, cout) if which == "fwd" else (cout, cin)
        line = f"{which:5s} CK{ck:3d}->CO{co:3d} k{k[1]} @{H}:"
        for v in variants:
            for tc in tchunks:
                if v == 0 and tc != tchunks[0]:
                    continue
                lib().vvae_conv3d_roll_config(v, tc)
                if which == "fwd":
                    out = ops.conv3d_fwd_raw(x, w, b).float(); ref = y0
                    f = lambda: ops.conv3d_fwd_raw(x, w, b)
                else:
                    out = ops.conv3d_dgrad_raw(dy, w).float(); ref = dx0
                    f = lambda: ops.conv3d_dgrad_raw(dy, w)
                err = ((out - ref).abs().max() / ref.abs().max()).item()
                t = tmg(f)
                line += f" | v{v}/t{tc} {t:6.1f}us {by / t / 1e3:5.0f}GB/s {fl / t / 1e6:4.0f}TF e{err:.0e}"
        print(line, flush=True)
lib().vvae_conv3d_roll_config(1, 0)
