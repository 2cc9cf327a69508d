"""Robustness soak of the graphed train step: bursts of unsynchronised steps (the host runs far ahead of the GPU, as in bench.py's timed region),
finiteness checked after each burst.   python tools/soak_bursts.py [bursts] [steps per burst]"""
import sys
sys.path.insert(0, ".")
import torch
import bench
import video_vae_amd as V
from video_vae_amd import loss as L, optim
from video_vae_amd.graph import GraphedTrainStep
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 40
import os
for kv in os.environ.get("HOOKS", "").split():                # HOOKS="vvae_gemm_nt_prefetch=0 ..." : tuning hooks set before the capture
    from video_vae_amd._lib import lib
    k, v = kv.split("=")
    assert getattr(lib(), k)(int(v)) == 0
sys.argv = [sys.argv[0], "--no-cpu-baseline"]
args = bench.parse()
dev = torch.device("cuda:0")
model, cfg = bench.build_model(args, dev, torch.bfloat16)
opt = optim.Optimizer(model, optim.reference_schedule(batch_size=args.batch))
g = torch.Generator().manual_seed(0)
video = torch.rand((args.batch, args.frames, args.size, args.size, 3), generator=g).to(dev, torch.bfloat16)
mask = torch.ones((args.batch, args.frames), device=dev)
step = GraphedTrainStep(model, opt, video, mask, L.HPARAMS, (args.size // cfg["patch_size"]) ** 2, V.Rngs(3))
for b in range(nb):
    for _ in range(ns):
        loss, aux = step()
    torch.cuda.synchronize()
    print(f"burst {b}: loss {float(loss):.4f} MSE {float(aux['MSE']):.4f}", flush=True)
    assert torch.isfinite(loss), "non-finite loss"
bad = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
assert not bad, bad[:5]
print("soak ok")
