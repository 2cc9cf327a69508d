import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from torch.profiler import profile, ProfilerActivity
sys.argv = ["bench.py"]
args = bench.parse()
import video_vae_amd as V
from video_vae_amd import ops, optim, loss as L
dev = torch.device("cuda", 0)
model, cfg = bench.build_model(args, dev, torch.bfloat16)
opt = optim.Optimizer(model, 1e-5)
g = torch.Generator().manual_seed(0)
video = torch.rand((4, 16, 256, 256, 3), generator=g).to(dev, torch.bfloat16); mask = torch.ones((4, 16), device=dev); rngs = V.Rngs(3)
def step():
    L.train_step(model, opt, video, mask, L.HPARAMS, 256, rngs)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=False) as prof:
    step()
    torch.cuda.synchronize()
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -e.count)
print("%-60s %6s %10s %10s" % ("name", "count", "cpu_us", "cuda_us"))
for e in rows[:0]:
    print("%-60s %6d %10.0f %10.0f" % (e.key[:60], e.count, e.cpu_time_total, e.device_time_total))
# which ops launch memsets: walk events, find memset device events and their parent cpu op
evs = prof.events()
from collections import Counter
cnt = Counter()
dur = {}
for ev in evs:
    if ev.device_type.name == "CPU" and ev.kernels:
        for k in ev.kernels:
            if any(t in k.name for t in ("fillBuffer", "Memset", "memset", "copyBuffer", "Memcpy", "elementwise_kernel", "vectorized_elementwise", "CatArray", "reduce_kernel")):
                p = ev
                names = []
                while p is not None and len(names) < 4:
                    names.append(p.name); p = p.cpu_parent
                key = (k.name[:70], " <- ".join(names))
                cnt[key] += 1
                dur[key] = dur.get(key, 0.0) + k.duration
for k, v in sorted(cnt.items(), key=lambda kv: -dur[kv[0]])[:45]:
    print(f"{v:4d} x {dur[k] / v:7.1f} us = {dur[k] / 1e3:6.3f} ms  {k[0]}  <-  {k[1]}")
