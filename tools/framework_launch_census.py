"""Which framework (aten) operators still launch kernels in one eager production train step, with their input shapes?

Every launch of the replayed hipGraph costs ~5 us however little it does; this lists what is left to fold into own kernels
(VERDICT r02 item 8).  Own C-ABI launches do not pass through the dispatcher and are not listed.

    python tools/framework_launch_census.py [bench.py flags]
"""
import collections
import sys

sys.path.insert(0, ".")
import torch
from torch.profiler import ProfilerActivity, profile

import bench
import video_vae_amd as V
from video_vae_amd import loss as L, optim

sys.argv = [sys.argv[0], "--no-cpu-baseline"] + sys.argv[1:]
args = bench.parse()
dev = torch.device("cuda:0")
model, cfg = bench.build_model(args, dev, torch.bfloat16)
opt = optim.Optimizer(model, optim.reference_schedule(batch_size=args.batch))
g = torch.Generator().manual_seed(0)
video = torch.rand((args.batch, args.frames, args.size, args.size, 3), generator=g).to(dev, torch.bfloat16)
mask = torch.ones((args.batch, args.frames), device=dev)
hw = (args.size // cfg["patch_size"]) ** 2
rngs = V.Rngs(3)
for _ in range(2):
    L.train_step(model, opt, video, mask, L.HPARAMS, hw, rngs)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    L.train_step(model, opt, video, mask, L.HPARAMS, hw, rngs)
    torch.cuda.synchronize()
seen = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and e.kernels:
        if not any(c.kernels for c in e.cpu_children):          # the leaf operator that launched
            seen[(e.name, str(e.input_shapes)[:150], tuple(k.name[:60] for k in e.kernels)[:2])] += 1
tot = 0
for (name, shapes, kern), n in sorted(seen.items(), key=lambda kv: (-kv[1], kv[0])):
    tot += n * len(kern)
    print(f"{n:3d} x {name:28s} {shapes}  -> {kern}")
print("framework launches per step:", tot)
