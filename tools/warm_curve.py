"""ms per step against time since the first step of a process: the chip takes a second or two of sustained load to settle on its clocks."""
import sys, time
sys.path.insert(0, ".")
import torch
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
step = GraphedTrainStep(model, opt, video, mask, L.HPARAMS, (args.size // cfg["patch_size"]) ** 2, V.Rngs(3))
torch.cuda.synchronize()
t00 = time.perf_counter()
out = []
for chunk in range(40):
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    out.append(f"{t1 - t00:5.2f}s {(t1 - t0) / 5 * 1e3:6.2f}")
print(" | ".join(out))
