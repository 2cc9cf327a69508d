"""Which framework op issues each device-to-device memcpy / copy kernel in the pass that GraphedTrainStep captures
(forward + autograd.grad backward + landing), profiled eagerly on the capture stream."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from torch.profiler import profile, ProfilerActivity
from collections import Counter
sys.argv = ["bench.py"]
args = bench.parse()
import video_vae_amd as V
from video_vae_amd import optim, loss as L
from video_vae_amd.graph import GraphedTrainStep
dev = torch.device("cuda", 0)
model, cfg = bench.build_model(args, dev, torch.bfloat16)
opt = optim.Optimizer(model, 1e-5)
video = torch.rand((4, 16, 256, 256, 3), generator=torch.Generator().manual_seed(0)).to(dev, torch.bfloat16)
mask = torch.ones((4, 16), device=dev)
g = GraphedTrainStep(model, opt, video, mask, L.HPARAMS, 256, V.Rngs(3))
with torch.cuda.stream(g.stream):
    opt.zero_grad(); g._fwd_bwd(); opt.update()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        opt.zero_grad(); g._fwd_bwd()
        torch.cuda.synchronize()
cnt, dur = Counter(), {}
for ev in prof.events():
    if ev.device_type.name == "CPU" and ev.kernels:
        for k in ev.kernels:
            if any(t in k.name for t in ("Memcpy", "copyBuffer", "Memset", "fillBuffer")):
                p, names = ev, []
                while p is not None and len(names) < 5:
                    names.append(p.name); p = p.cpu_parent
                key = (k.name[:40], " <- ".join(names))
                cnt[key] += 1; dur[key] = dur.get(key, 0.0) + k.duration
for k, v in sorted(cnt.items(), key=lambda kv: -dur[kv[0]])[:25]:
    print(f"{v:4d} x {dur[k] / v:7.1f} us = {dur[k] / 1e3:6.3f} ms  {k[0]}  <-  {k[1]}")
