"""A/B of a tuning hook on the whole train step in ONE process on ONE box (box-to-box spread is larger than most kernel-level gains):
captures the production step once per hook value (kernel arguments are baked at capture) and alternates timed runs of the two graphs.
    python tools/ab_hook.py vvae_gemm_nt_stagger 0 2 [r=rounds]             (a C hook; any number of values)
    python tools/ab_hook.py vvae_conv3d_roll_config 1,0 1,8 1,4                (a two-argument hook)
    python tools/ab_hook.py py:video_vae_amd.layers.NT_SILU 0 1                (a module-level switch)"""
import sys, time
sys.path.insert(0, ".")
import torch
import bench
import video_vae_amd as V
from video_vae_amd import loss as L, optim
from video_vae_amd.graph import GraphedTrainStep
from video_vae_amd._lib import lib

hook = sys.argv[1]
vals = [tuple(int(x) for x in v.split(",")) for v in sys.argv[2:] if not v.startswith("r=")]      # 1,8 = a two-argument hook call
rounds = next((int(v[2:]) for v in sys.argv[2:] if v.startswith("r=")), 4)
sys.argv = [sys.argv[0], "--no-cpu-baseline"]
args = bench.parse()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
model, cfg = bench.build_model(args, dev, torch.bfloat16)
opt = optim.Optimizer(model, optim.reference_schedule(batch_size=args.batch))
g = torch.Generator().manual_seed(0)
video = torch.rand((args.batch, args.frames, args.size, args.size, 3), generator=g).to(dev, torch.bfloat16)
mask = torch.ones((args.batch, args.frames), device=dev)
hw = (args.size // cfg["patch_size"]) ** 2
steps = {}
def set_value(v):
    if hook.startswith("py:"):                            # a module-level switch: py:video_vae_amd.layers.NT_SILU
        import importlib
        mod, name = hook[3:].rsplit(".", 1)
        m = importlib.import_module(mod)
        if isinstance(getattr(m, name), list):               # a one-element list switch (ops.GN_POOL_FWD_FUSED = [True])
            getattr(m, name)[0] = type(getattr(m, name)[0])(v[0])
        else:
            setattr(m, name, v[0])
    else:
        assert getattr(lib(), hook)(*v) == 0


for v in vals:
    set_value(v)
    steps[v] = GraphedTrainStep(model, opt, video, mask, L.HPARAMS, hw, V.Rngs(3))
    for _ in range(3):
        steps[v]()
torch.cuda.synchronize()
res = {v: [] for v in vals}
for r in range(rounds):
    for v in vals:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            steps[v]()
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t0) / 10 * 1e3)
for v in vals:
    print(f"{hook}{v}: " + " ".join(f"{t:.2f}" for t in res[v]) + f"  ms/step, median {sorted(res[v])[len(res[v]) // 2]:.2f}", flush=True)
