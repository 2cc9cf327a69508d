"""Is one graphed production train step a pure function of its inputs?  The same step (same parameters, moments, noise) is replayed after different
histories -- straight away, after a host pause, after a burst of 17 other steps (hot clocks, different cache contents) -- and the flat gradient
buffer and the loss are compared bit for bit with the first replay; a slot that differs names the kernel that wrote it.
    [BENCH_ARGS="--batch 2 --frames 32"] python tools/step_determinism.py [trials] [eager]"""
import sys, time
sys.path.insert(0, ".")
import torch
import bench
import video_vae_amd as V
from video_vae_amd import loss as L, optim
from video_vae_amd.graph import GraphedTrainStep
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 6
eager = len(sys.argv) > 2 and sys.argv[2] == "eager"
import os
sys.argv = [sys.argv[0], "--no-cpu-baseline"] + os.environ.get("BENCH_ARGS", "").split()      # e.g. BENCH_ARGS="--batch 2 --frames 32"
args = bench.parse()
dev = torch.device("cuda:0")
model, cfg = bench.build_model(args, dev, torch.bfloat16)
opt = optim.Optimizer(model, optim.reference_schedule(batch_size=args.batch))
g = torch.Generator().manual_seed(0)
video = torch.rand((args.batch, args.frames, args.size, args.size, 3), generator=g).to(dev, torch.bfloat16)
mask = torch.ones((args.batch, args.frames), device=dev)
hw = (args.size // cfg["patch_size"]) ** 2
if eager:
    class Eager:
        aux = {}
        def __init__(self): self.rngs = V.Rngs(3)
        def __call__(self):
            out = L.train_step(model, opt, video, mask, L.HPARAMS, hw, self.rngs)
            self.aux = {k: v for k, v in out[1].items() if k != "reconstruction"}
            return out
    step = Eager()
else:
    step = GraphedTrainStep(model, opt, video, mask, L.HPARAMS, hw, V.Rngs(3))
for _ in range(5):
    step()
torch.cuda.synchronize()
import copy
state = (opt.p.clone(), opt.m.clone(), opt.v.clone(), opt.count, copy.deepcopy(step.rngs.counts) if eager else step.gen.get_state())


def restore():
    opt.p.copy_(state[0]); opt.m.copy_(state[1]); opt.v.copy_(state[2]); opt.count = state[3]
    opt.refresh_shadow()
    if eager:
        step.rngs.counts = copy.deepcopy(state[4])
    else:
        step.gen.set_state(state[4])


def once(history):
    restore()
    history()
    restore()
    torch.cuda.synchronize()
    loss, aux = step()
    torch.cuda.synchronize()
    return float(loss), opt.g.clone(), opt.p.clone()


def burst():
    for _ in range(17):
        step()


histories = {"none": lambda: None, "pause": lambda: time.sleep(0.05), "burst17": burst,
             "burst17+sync": lambda: (burst(), torch.cuda.synchronize()), "reads": lambda: [float(v) for v in step.aux.values()]}
l0, g0, p0 = once(histories["none"])
names = [n for n, _ in model.named_parameters()]
bad_total = 0
for t in range(trials):
    for hn, h in histories.items():
        l, gg, pp = once(h)
        if l != l0 or not torch.equal(gg, g0) or not torch.equal(pp, p0):
            bad_total += 1
            slots = [n for n, prm, o in zip(opt.names, opt.params, opt.offsets) if not torch.equal(gg[o:o + prm.numel()], g0[o:o + prm.numel()])]
            nd = int((gg != g0).sum())
            print(f"trial {t} history {hn}: loss {l!r} vs {l0!r}; {nd} gradient elements differ, params differ {not torch.equal(pp, p0)}; slots {slots[:8]}", flush=True)
print("differences:", bad_total, "of", trials * len(histories))
