"""Which copies (hipMemcpyAsync / aten::copy_) does one GRAPHED production step issue outside the replayed graph, and from where?

    python tools/step_memcpy_probe.py
"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

import bench
import video_vae_amd as V
from video_vae_amd import loss as L, optim
from video_vae_amd.graph import GraphedTrainStep

sys.argv = [sys.argv[0], "--no-cpu-baseline"]
args = bench.parse()
dev = torch.device("cuda:0")
model, cfg = bench.build_model(args, dev, torch.bfloat16)
opt = optim.Optimizer(model, optim.reference_schedule(batch_size=args.batch))
g = torch.Generator().manual_seed(0)
video = torch.rand((args.batch, args.frames, args.size, args.size, 3), generator=g).to(dev, torch.bfloat16)
mask = torch.ones((args.batch, args.frames), device=dev)
hw = (args.size // cfg["patch_size"]) ** 2
step = GraphedTrainStep(model, opt, video, mask, L.HPARAMS, hw, V.Rngs(3))
for _ in range(3):
    step()
torch.cuda.synchronize()
N = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for _ in range(N):
        step()
    torch.cuda.synchronize()
seen = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.kernels and not any(c.kernels for c in e.cpu_children):
        names = tuple(k.name[:50] for k in e.kernels)
        if any("emcpy" in n or "copyBuffer" in n or "fill" in n.lower() for n in names) or e.name.startswith("aten::"):
            st = [s for s in (e.stack or []) if "video_vae_amd" in s or "bench.py" in s][:2]
            seen[(e.name, str(e.input_shapes)[:80], names[:1], tuple(s.split("/")[-1][:60] for s in st))] += 1
for k, n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(f"{n / N:5.1f} per step  {k[0]:24s} {k[1]:80s} {k[2]} {k[3]}")
