import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
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
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(38); print(s.getvalue()[:7000])
