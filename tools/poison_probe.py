"""Uninitialised-read hunt: every torch.empty / empty_like / new_empty of the process is poisoned (NaN for floating types, 0x7f bytes
otherwise) before a few eager production train steps; a kernel that reads memory nobody wrote turns the loss or a gradient non-finite
at once instead of depending on what the allocator happened to hand back.   python tools/poison_probe.py [steps]"""
import sys
sys.path.insert(0, ".")
import torch

_empty, _empty_like = torch.empty, torch.empty_like


def _poison(t):
    if t.numel():
        if t.is_floating_point():
            t.fill_(float("nan"))
        elif t.dtype in (torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64):
            t.fill_(0x7f)
    return t


def empty(*a, **k):
    return _poison(_empty(*a, **k))


def empty_like(*a, **k):
    return _poison(_empty_like(*a, **k))


torch.empty, torch.empty_like = empty, empty_like
_new_empty = torch.Tensor.new_empty
torch.Tensor.new_empty = lambda self, *a, **k: _poison(_new_empty(self, *a, **k))

import bench
import video_vae_amd as V
from video_vae_amd import loss as L, optim

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sys.argv = [sys.argv[0], "--no-cpu-baseline"]
args = bench.parse()
dev = torch.device("cuda:0")
model, cfg = bench.build_model(args, dev, torch.bfloat16)
opt = optim.Optimizer(model, optim.reference_schedule(batch_size=args.batch))
g = torch.Generator().manual_seed(0)
video = torch.rand((args.batch, args.frames, args.size, args.size, 3), generator=g).to(dev, torch.bfloat16)
mask = torch.ones((args.batch, args.frames), device=dev)
mask[1, 12:] = 0                                              # a masked tail too
hw = (args.size // cfg["patch_size"]) ** 2
rngs = V.Rngs(3)
for i in range(steps):
    loss, aux = L.train_step(model, opt, video, mask, L.HPARAMS, hw, rngs)
    ok = bool(torch.isfinite(loss)) and bool(torch.isfinite(opt.g).all()) and bool(torch.isfinite(opt.p).all())
    print(f"step {i}: loss {float(loss):.5f} finite grads {bool(torch.isfinite(opt.g).all())} finite params {bool(torch.isfinite(opt.p).all())}", flush=True)
    if not ok:
        bad = [n for n, p in model.named_parameters() if getattr(p, 'gview', None) is not None and not torch.isfinite(p.gview).all()]
        print("non-finite gradient slots:", bad[:12], len(bad))
        sys.exit(1)
print("poison probe ok")
