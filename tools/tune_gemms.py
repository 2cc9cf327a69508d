#!/usr/bin/env python3
"""Offline selection of library GEMM solutions for the trunk's Linear products (PyTorch TunableOp over hipBLASLt / rocBLAS).

    python tools/tune_gemms.py <out.csv>

Runs two EAGER train steps of the production model (C3 shapes) with tuning on, so every distinct (transA, transB, m, n, k, ld)
product of the step is timed once against the library's candidate solutions, and writes the winners to <out.csv>.  The shipped
run never tunes: it may load such a file with tuning OFF (lookup only)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.abspath(sys.argv[1])
import torch.cuda.tunable as tn  # noqa: E402
tn.enable(True)
tn.tuning_enable(True)
tn.set_max_tuning_duration(30)
tn.set_max_tuning_iterations(100)
tn.set_filename(out, insert_device_ordinal=False)

import video_vae_amd as V  # noqa: E402
from video_vae_amd import loss as L, optim  # noqa: E402

dev = torch.device("cuda", 0)
cfg = dict(height=256, width=256, channels=3, patch_size=16, encoder_depth=2, decoder_depth=2, mlp_dim=1536, num_heads=8,
           qkv_features=512, max_temporal_len=64, spatial_compression_rate=8, unembedding_upsample_rate=4)      # 2 + 2 layers: every shape
m = V.VideoVAE(rngs=V.Rngs(2), dtype=torch.bfloat16, **cfg).to(dev)
with torch.no_grad():
    fc = m.decoder.unet.final_conv
    fc.kernel.copy_(torch.randn(fc.kernel.shape) * 0.2)
opt = optim.Optimizer(m, 1e-4)
video = torch.rand((4, 16, 256, 256, 3)).to(dev, torch.bfloat16)
mask = torch.ones((4, 16), device=dev)
t0 = time.time()
for i in range(2):
    loss, _ = L.train_step(m, opt, video, mask, L.HPARAMS, 256, V.Rngs(3))
    torch.cuda.synchronize()
    print(f"step {i}: loss {float(loss):.4f}, {time.time() - t0:.1f} s", flush=True)
print('results so far:', len(tn.get_results()), 'entries; the file is written at exit', flush=True)
