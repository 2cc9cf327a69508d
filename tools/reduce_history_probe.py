"""What exactly goes wrong with the framework's multi-block reduction inside the replayed production step?

Puts the three `dy.sum(0)` bias-gradient reductions of the 96-wide latent heads back (layers.FRAMEWORK_COLSUM), captures the production
train step, and replays ONE step from the same parameters / moments / noise after different histories (as tools/step_determinism.py).
For every replay whose gradient buffer differs from the first one it reports, per differing slot:
  * how many elements differ, and whether the wrong elements equal, bit for bit, the values the SAME slot held after the replay that ran
    just before (the history's last step, which had other noise) -- "stale output": the reduction's last workgroup never wrote;
  * whether every wrong element is explained as this step's sum with k of the reduction's partial rows taken from the previous replay
    (cannot be decided from outside the kernel; reported as "neither" when the value matches neither the reference nor the stale value).
It also writes the captured graph as DOT (hipGraphDebugDotPrint) and lists the memset nodes with their in / out edges.

    python tools/reduce_history_probe.py [trials] [dot path]
"""
import os
import re
import sys
import time

sys.path.insert(0, ".")
import torch

import bench
import video_vae_amd as V
from video_vae_amd import layers as LY, loss as L, optim
from video_vae_amd.graph import GraphedTrainStep

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dot = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/step_graph.dot"
framework = os.environ.get("PROBE_FRAMEWORK_SUM", "1") == "1"
sys.argv = [sys.argv[0], "--no-cpu-baseline"]
args = bench.parse()
dev = torch.device("cuda:0")
LY.FRAMEWORK_COLSUM[0] = framework
model, cfg = bench.build_model(args, dev, torch.bfloat16)
opt = optim.Optimizer(model, optim.reference_schedule(batch_size=args.batch))
g = torch.Generator().manual_seed(0)
video = torch.rand((args.batch, args.frames, args.size, args.size, 3), generator=g).to(dev, torch.bfloat16)
mask = torch.ones((args.batch, args.frames), device=dev)
hw = (args.size // cfg["patch_size"]) ** 2
os.makedirs(os.path.dirname(dot) or ".", exist_ok=True)
step = GraphedTrainStep(model, opt, video, mask, L.HPARAMS, hw, V.Rngs(3), debug_dot=dot)

# ---- the captured graph: memset nodes and their edges
try:
    txt = open(dot).read()
    nodes = dict(re.findall(r'"?(\w+)"?\s*\[[^\]]*label="([^"]*)"', txt))
    edges = re.findall(r'"?(\w+)"?\s*->\s*"?(\w+)"?', txt)
    mem = [n for n, lab in nodes.items() if "MEMSET" in lab.upper() or "memset" in lab]
    print(f"graph: {len(nodes)} nodes, {len(edges)} edges, {len(mem)} memset nodes", flush=True)
    for n in mem[:12]:
        ins = [nodes.get(a, a)[:60] for a, b in edges if b == n]
        outs = [nodes.get(b, b)[:60] for a, b in edges if a == n]
        print(f"  memset {n}: {nodes[n][:80]!r}\n     in  <- {ins}\n     out -> {outs}", flush=True)
except Exception as e:                                                  # the dump is evidence, not a requirement
    print(f"graph dump unreadable: {type(e).__name__}: {e}", flush=True)

for _ in range(5):
    step()
torch.cuda.synchronize()
state = (opt.p.clone(), opt.m.clone(), opt.v.clone(), opt.count, step.gen.get_state())


def restore():
    opt.p.copy_(state[0]); opt.m.copy_(state[1]); opt.v.copy_(state[2]); opt.count = state[3]
    opt.refresh_shadow()
    step.gen.set_state(state[4])


def once(history):
    restore()
    history()
    torch.cuda.synchronize()
    g_prev = opt.g.clone()                                              # what every slot held after the replay just before
    restore()
    torch.cuda.synchronize()
    loss, aux = step()
    torch.cuda.synchronize()
    return float(loss), opt.g.clone(), g_prev


def burst(n):
    def run():
        for _ in range(n):
            step()
    return run


histories = {"none": lambda: None, "pause": lambda: time.sleep(0.05), "burst1": burst(1), "burst17": burst(17),
             "burst17+sync": lambda: (burst(17)(), torch.cuda.synchronize())}
l0, g0, _ = once(histories["burst1"])
l1, g1, _ = once(histories["burst1"])
print(f"framework sums {'ON' if framework else 'off'}; reference replay twice: loss equal {l0 == l1}, gradients equal {torch.equal(g0, g1)}", flush=True)
bad_total = 0
for t in range(trials):
    for hn, h in histories.items():
        l, gg, gp = once(h)
        if l != l0 or not torch.equal(gg, g0):
            bad_total += 1
            print(f"trial {t} history {hn}: loss {l!r} vs {l0!r}", flush=True)
            for n, prm, o in zip(opt.names, opt.params, opt.offsets):
                a, r, p = gg[o:o + prm.numel()], g0[o:o + prm.numel()], gp[o:o + prm.numel()]
                d = a != r
                if d.any():
                    nd = int(d.sum())
                    stale = int((a[d] == p[d]).sum())
                    relerr = float((a[d] - r[d]).abs().max() / r.abs().max().clamp_min(1e-30))
                    print(f"    {n}: {nd} of {prm.numel()} elements differ; {stale} of them hold the previous replay's value bit for bit; "
                          f"max |diff| / max |ref| = {relerr:.3e}; finite {bool(torch.isfinite(a).all())}", flush=True)
print("differences:", bad_total, "of", trials * len(histories))
