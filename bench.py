#!/usr/bin/env python3
"""bench.py -- video frames/sec (fwd+bwd+optimizer) of the video-VAE training step on N MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 runs one rank per GPU over RCCL.  Under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (the
driver's form) this process IS a rank.  Started bare (`python bench.py --gpus N`, WORLD_SIZE unset) it launches the ranks itself:
a child `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` from a parent that never
touches the GPU (no exec after a GPU call), relays rank 0's JSON line and exits non-zero if any rank failed -- the counterpart
of the reference's self-initialising ranks (claude_distributed/distributed_train.py:79, distributed_run.sh:1-9).

A "step" is one pass of the hot path over one synthetic batch: VideoVAE forward (Encoder -> reparameterise -> Decoder with
the 3D-conv UNet), masked recon+KL loss, backward, clip-by-global-norm + Adam.  Workload at every N = BASELINE.json
config C3 per GPU: B=4 clips of 3x16x256x256 (written (B,T,H,W,C) = (4,16,256,256,3)), bf16 compute, fp32 parameters, the
reference's production model (train/rl_nonadversarial.py:234-236) in the train/model.py flavour.  Weak scaling: per-GPU
batch fixed, gradients all-reduced over RCCL overlapped with backward (video_vae_amd/ddp.py).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the north-star Conv3d kernel: algorithmic bytes and FLOPs per launch / HIP-event duration measured inside the timed steps, against
                  the lower of the HBM and the matrix roof for that launch (`frac`; the HBM fraction is always there as `frac_hbm`);
  cpu_baseline -- the CPU oracle (oracle/, a port of the reference's math) timed on the host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import socket  # noqa: E402
import subprocess  # noqa: E402

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFS = 2500.0         # MI355X_MICROARCH.md: dense bf16 MFMA peak (AMD's 5 PF figure is 2:1 sparse)
PROD = dict(height=256, width=256, channels=3, patch_size=16, encoder_depth=9, decoder_depth=12, mlp_dim=1536, num_heads=8,
            qkv_features=512, max_temporal_len=64, spatial_compression_rate=8, unembedding_upsample_rate=4)


TRAFFIC_FILE = "r04_traffic.json"


def measured_traffic(kernel, key="hbm_bytes_per_launch"):
    """HBM bytes per launch of ``kernel`` (or, with key="mfma_busy", the busy fraction of the matrix pipes during its launches) from
    the committed rocprofv3 PMC passes (profiles/r04_traffic.json: separate FETCH_SIZE and WRITE_SIZE passes, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950; a third pass with SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE) -- or None, with the
    reason, when there is no figure or the kernel's sources have changed since the passes were taken (the file carries the
    sha256 of the .hip files the kernel is built from: a stale counter figure is refused, not quoted)."""
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    try:
        with open(os.path.join(here, "profiles", TRAFFIC_FILE)) as f:
            rec = json.load(f).get(kernel)
    except (OSError, ValueError):
        return None, f"profiles/{TRAFFIC_FILE} missing"
    if not rec or key not in rec:
        return None, f"no PMC figure ({key}) for {kernel} in profiles/{TRAFFIC_FILE}"
    h = hashlib.sha256()
    for name in rec.get("source_files", []):
        try:
            with open(os.path.join(here, "video_vae_amd", "csrc", name), "rb") as f:
                h.update(f.read())
        except OSError:
            return None, f"{name} not found"
    if rec.get("source_sha256") != h.hexdigest():
        return None, f"PMC passes predate the current {'/'.join(rec.get('source_files', []))}: re-run tools/pmc_bench_sum.py"
    return rec.get(key), None


def source_sha256(files):
    import hashlib
    h = hashlib.sha256()
    for name in files:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "video_vae_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--settle-seconds", type=float, default=1.0,
                    help="untimed steps run for this long before the W warm-up steps: from a cold start the chip needs ~0.4 s of sustained load "
                         "to settle on its clocks (35.2 -> 32.9 ms per step, tools/warm_curve.py); 0 = none")
    ap.add_argument("--batch", type=int, default=4, help="clips per GPU")
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--workload", default="vae", choices=["vae", "unet"],
                    help="vae: full VideoVAE train step (the metric); unet: the Conv3d UNet stack alone (diagnostic)")
    ap.add_argument("--flavour", default="model", choices=["model", "rl"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-ddp", action="store_true",
                    help="attach the gradient reducer and a process group even at world size 1 (launch under torch.distributed.run)")
    ap.add_argument("--split-graph", action="store_true",
                    help="capture the step as two graphs cut at the encoder's last block even at N = 1 (the N > 1 default)")
    ap.add_argument("--grad-dtype", default="f32", choices=["f32", "bf16"],
                    help="dtype the gradient buckets cross the transport in (bf16: half the xGMI bytes, see video_vae_amd/ddp.py); f32 = the reference")
    ap.add_argument("--enc-segments", type=int, default=3,
                    help="data parallel, graph mode: hipGraphs the encoder's backward is cut into (1 + this many graphs per step).  The buckets of a "
                         "segment are all-reduced under the segments still to come; only the LAST segment's wait for the end of backward: 3 -> 95 MB "
                         "of the 683 MB, 9 -> one encoder block + the patch embedding = 32 MB.  Every cut costs ~0.2 ms of step time at N = 1 "
                         "(profiles/r03_bench_force_ddp_line.json: 9 segments 37.0 ms, 3 segments 35.7, one graph 34.9), more than a ring moves 63 MB "
                         "in: the default stays 3")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the measured configuration); gloo only to rehearse N > 1 on a one-GPU box")
    ap.add_argument("--no-graph", "--eager", action="store_true", dest="no_graph",
                    help="run the timed steps eagerly (default: forward+backward replayed from a captured hipGraph)")
    ap.add_argument("--with-input-pipeline", action="store_true",
                    help="feed every timed step from the host input pipeline (video_vae_amd/data.py: worker processes -> pinned uint8 "
                         "-> H2D on a side stream) instead of a batch resident in HBM; the line reports the same metric")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="skip the per-launch HIP-event leg (roofline / kernels / conv_stack): for profiler passes over this script, where "
                         "every extra step is minutes of serialised dispatches")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the secondary block `also` (short runs of the reference driver's own flavour -- rl_model -- and of config C5's per-GPU "
                         "shape, after the headline's timed region)")
    ap.add_argument("--also-steps", type=int, default=8, help="timed steps of each secondary run")
    ap.add_argument("--cpu-frames", type=int, default=4, help="frames per clip of the bounded CPU sample")
    ap.add_argument("--cpu-clips", type=int, default=1, help="clips of the bounded CPU sample")
    ap.add_argument("--cpu-budget", type=float, default=60.0, help="seconds the CPU baseline leg may take (steps are cut to fit)")
    return ap.parse_args()


def build_model(args, dev, dtype):
    import video_vae_amd as V
    from video_vae_amd import rl_model
    cfg = dict(PROD, height=args.size, width=args.size)
    rngs = V.Rngs(2)                                            # nnx.Rngs(2), rl_nonadversarial.py:236
    if args.workload == "unet":
        m = V.UNet(channels=12, base_features=16, num_levels=3, out_features=3, rngs=rngs, dtype=dtype)
        fc = m.final_conv
    else:
        cls = rl_model.VideoVAE if args.flavour == "rl" else V.VideoVAE
        m = cls(rngs=rngs, dtype=dtype, **cfg)
        fc = m.decoder.unet.final_conv
    # final_conv is zero-initialised in the reference (unet.py:150), which makes the UNet a no-op and blocks its
    # interior gradients; the bench re-initialises it non-zero so the conv stack does real work.
    with torch.no_grad():
        g = torch.Generator().manual_seed(1234)
        fc.kernel.copy_(torch.randn(fc.kernel.shape, generator=g) * (fc.kernel.shape[-2] ** -0.5))
    return m.to(dev), cfg


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_cores():
    """Cores this process may actually use: the affinity mask clipped by the cgroup CPU quota (a GPU box hands one GPU's share of
    a large host to the job: os.cpu_count() reports the whole host there and oversubscribes the share many times over)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                       # cgroup v2: "<quota|max> <period>"
            q, per = f.read().split()[:2]
            if q != "max":
                quota = int(q) / int(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:   # cgroup v1
                q, per = int(f.read()), int(g.read())
                if q > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota + 0.5)))
    env = os.environ.get("VVAE_CPU_THREADS")
    return max(1, int(env)) if env else n


def cpu_baseline(args):
    """the CPU oracle (a port of the reference's math: PyTorch CPU, fp32, oneDNN) on all host cores this job may
    use, on a bounded sample of the same workload: 3 warm-up + 5 timed steps, median.  ``value`` is the same metric as the GPU
    line -- frames/s of the whole train step (full-depth VideoVAE forward + recon/KL loss + backward + clip + Adam) -- on a clip
    cut down in batch and frames; the Conv3d UNet alone is timed beside it.  The leg is bounded in TIME as well: when the first
    step shows that 3 + 5 steps would not fit ``--cpu-budget`` seconds, fewer are run and the protocol field says how many."""
    import statistics
    from oracle import loss as OLoss, model as OM, optim as OOpt, unet as OU
    threads = host_cores()
    torch.set_num_threads(threads)
    s, t, nb = args.size, args.cpu_frames, args.cpu_clips
    g = torch.Generator().manual_seed(0)

    def timed(fn, budget):
        t0 = time.perf_counter()
        fn()
        first = time.perf_counter() - t0                     # warm-up 1 (also pays one-time allocations)
        warm, reps = 2, 5
        if first * (warm + reps) > budget:
            warm = 0 if first * 4 > budget else 1
            reps = max(1, min(5, int(budget / max(first, 1e-3)) - warm))
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts), f"{warm + 1} warm-up + {reps} timed"

    # ---- whole train step, production depth
    cfg = OM.VAEConfig(**dict(PROD, height=s, width=s))
    p = OM.init_video_vae(cfg, seed=2, zero_final=False)
    adam = OOpt.Adam(p)
    video = torch.rand((nb, t, s, s, 3), generator=g)
    mask = torch.ones(nb, t)
    emask = OLoss.expand_mask(mask.bool(), cfg.hw)
    noise = {"gumbel_u": torch.rand((nb, t, 1), generator=g), "reparam_eps": torch.randn((nb, t, cfg.hw, cfg.latent_dim), generator=g)}
    state = {"p": p}

    def vae_step():
        pr = {k: v.detach().requires_grad_(True) for k, v in state["p"].items()}
        loss, _aux = OLoss.loss_fn_plain(OM.video_vae(pr, cfg, video, emask, noise), video, mask)
        loss.backward()
        clipped, _gn = OOpt.clip_by_global_norm({k: v.grad for k, v in pr.items()}, 1.0)
        state["p"] = adam.update({k: v.detach() for k, v in pr.items()}, clipped, 2e-5)
    t_vae, proto_vae = timed(vae_step, 0.75 * args.cpu_budget)
    del p, adam, state
    # ---- the Conv3d UNet alone (the north-star kernels' CPU counterpart)
    pu = OU.init_unet(12, 16, 3, 3, seed=5, zero_final=False)
    for v in pu.values():
        v.requires_grad_(True)
    x = torch.randn((nb, t, s, s, 12), generator=g) * 0.5

    def unet_step():
        for v in pu.values():
            v.grad = None
        OU.unet(pu, x).square().mean().backward()
    t_unet, proto_unet = timed(unet_step, 0.25 * args.cpu_budget)
    return {"value": nb * t / t_vae, "unit": "frames/s", "cores": threads, "cpu_model": cpu_model_name(), "kind": "port",
            "host_logical_cpus": os.cpu_count(),
            "protocol": f"{proto_vae} steps, median; torch.set_num_threads({threads}) = affinity mask clipped by the cgroup CPU quota",
            "sample": f"oracle full-depth VideoVAE train step (fwd + recon/KL loss + bwd + clip + Adam), fp32, {nb} clip(s) x {t} frames x "
                      f"{s}x{s}x3, median {t_vae:.2f} s/step on {threads} threads",
            "unet_only": {"value": nb * t / t_unet, "unit": "frames/s", "protocol": f"{proto_unet} steps, median",
                          "sample": f"oracle UNet fwd+bwd, fp32, {nb} clip(s) x {t} frames x {s}x{s}x12 features, median {t_unet:.2f} s/step"}}


def also_runs(args, dev, dtype):
    """The secondary block of the JSON line: what the headline does not show but the reference driver runs.
      rl_flavour -- train/rl_nonadversarial.py:6 builds rl_model.VideoVAE (pair-doubled decoder, Bernoulli frame masks, REINFORCE term), not
                    the model.py flavour the metric is quoted on: same shape as the headline, the rl loss;
      c5_shape   -- BASELINE.json configs[4] per GPU: B=2 clips of 3x32x256x256 (temporal attention over T=32), model.py flavour.
    Short runs (2 warm-up + --also-steps timed graph replays, median of per-step HIP events), started only after the headline's timed
    region and kernel-timing leg are over and its model is freed, so they cannot perturb it.  A failure is reported, never raised."""
    import copy
    import gc
    import statistics
    import video_vae_amd as V
    from video_vae_amd import loss as L, optim
    from video_vae_amd.graph import GraphedTrainStep
    out = {}
    for name, flavour, B, T in (("rl_flavour", "rl", args.batch, args.frames), ("c5_shape", "model", 2, 32)):
        try:
            a = copy.copy(args)
            a.flavour, a.batch, a.frames, a.workload = flavour, B, T, "vae"
            model, cfg = build_model(a, dev, dtype)
            opt = optim.Optimizer(model, optim.reference_schedule(batch_size=B))
            g = torch.Generator().manual_seed(0)
            video = torch.rand((B, T, a.size, a.size, 3), generator=g).to(dev, dtype)
            mask = torch.ones((B, T), device=dev)
            step = GraphedTrainStep(model, opt, video, mask, L.HPARAMS, (a.size // cfg["patch_size"]) ** 2, V.Rngs(3), warmup=1)
            for _ in range(2):
                loss, _aux = step()
            torch.cuda.synchronize()
            k = max(1, args.also_steps)
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
            for i in range(k):
                evs[i].record()
                loss, _aux = step()
            evs[k].record()
            torch.cuda.synchronize()
            ms = statistics.median(evs[i].elapsed_time(evs[i + 1]) for i in range(k))
            out[name] = {"ms_per_step": ms, "frames_per_s": B * T / (ms * 1e-3), "graph_nodes": step.census, "steps": k, "finite": bool(torch.isfinite(loss)),
                         "workload": f"{'rl_model' if flavour == 'rl' else 'model'}.VideoVAE train step, B={B} x 3x{T}x{a.size}x{a.size}, {args.dtype}, graph replay"}
            del step, opt, model, video, mask, loss, _aux
        except Exception as e:                                   # secondary figures: never take the headline down with them
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        gc.collect()
        torch.cuda.empty_cache()
    return out


def self_launch(n):
    """Parent of a bare `bench.py --gpus N`: start N ranks under torch.distributed.run, relay rank 0's JSON line.

    This process has not initialised the GPU (importing torch does not) and never will; the ranks are child processes."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:                                     # ranks' stderr passes straight through
        if out.lstrip().startswith("{") and '"metric"' in out:
            line = out.strip()
        else:
            sys.stdout.write(out)
    rc = proc.wait()
    if rc != 0:
        sys.exit(f"bench.py: a rank failed (torch.distributed.run exit code {rc})" if rc > 0 else f"bench.py: launcher killed by signal {-rc}")
    if line is None:
        sys.exit("bench.py: the ranks finished without printing a result line")
    print(line, flush=True)


def main():
    args = parse()
    # dmabuf IPC: RCCL across processes needs it on this pool (the host driver has no legacy IPC).  Set before the first GPU call of
    # THIS process too -- under the driver's `torch.distributed.run ... bench.py` form no self-launch parent exists to export it.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with --nproc-per-node {args.gpus} (or bare, without WORLD_SIZE)")
    ddp_on = world > 1 or args.force_ddp     # --force-ddp: the whole process-group path with a single rank (RCCL rehearsal)
    if args.backend != "nccl":               # rehearsal of the N > 1 path on a box with fewer GPUs than ranks (gloo transport)
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if ddp_on:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32

    import video_vae_amd as V
    from video_vae_amd import ops, optim, ddp, loss as L
    model, cfg = build_model(args, dev, dtype)
    opt = optim.Optimizer(model, optim.reference_schedule(batch_size=args.batch * world))
    reducer = None
    if ddp_on:
        reducer = ddp.GradReducer(opt, grad_dtype=torch.bfloat16 if args.grad_dtype == "bf16" else torch.float32)
        reducer.broadcast_parameters(0)
    nparams = sum(p.numel() for p in model.parameters())

    B, T, S = args.batch, args.frames, args.size
    g = torch.Generator().manual_seed(0 + rank)                 # per-rank data seed = seed + rank
    rngs = V.Rngs(3 + rank)
    mode = "eager"
    if args.workload == "unet":
        feat = (torch.randn((B, T, S, S, 12), generator=g) * 0.5).to(dev, dtype)
        tgt = torch.rand((B, T, S, S, 3), generator=g).to(dev, dtype)
        mask = torch.ones((B, T), device=dev)

        def step():
            opt.zero_grad()
            recon = model(feat)
            mse, _ = ops.masked_mse_mae(tgt, recon, mask, 1)
            loss = mse.mean()
            loss.backward()
            opt.update()
            return loss
    else:
        video = torch.rand((B, T, S, S, 3), generator=g).to(dev, dtype)      # dataloader range [0,1), cast as :330
        mask = torch.ones((B, T), device=dev)
        hw = (S // cfg["patch_size"]) ** 2

        feed = None
        if args.with_input_pipeline:
            # every step's batch comes off disk through worker processes, pinned uint8 staging and an H2D copy on a side stream
            import tempfile
            from video_vae_amd import data as D
            clip_dir = tempfile.mkdtemp(prefix=f"vvae_bench_clips_r{rank}_")
            D.write_synthetic_clips(clip_dir, max(48, 4 * B), T + 4, S + 32, S + 32, seed=rank)
            workers = max(2, min(8, host_cores() // max(1, world)))
            host = D.create_batched_dataloader(clip_dir, batch_size=B, max_frames=T, resize=(S, S), crop_size=S, shuffle=True, seed=0,
                                               num_workers=workers, prefetch_size=4 * workers, drop_remainder=True, rank=rank,
                                               num_epochs=None, as_uint8=True)
            feed = D.DevicePrefetcher(host, dev, dtype=dtype)

        def batch():
            if feed is None:
                return video, mask
            b = next(feed)
            return b["video"], b["mask"]

        def eager_step():
            v, m = batch()
            loss, _aux = L.train_step(model, opt, v, m, L.HPARAMS, hw, rngs)
            return loss

        step, mode, graph_nodes = eager_step, "eager", None
        if not args.no_graph:
            try:
                from video_vae_amd.graph import GraphedTrainStep
                gstep = GraphedTrainStep(model, opt, video, mask, L.HPARAMS, hw, rngs, split=True if args.split_graph else None,
                                         enc_segments=args.enc_segments)

                def step():
                    loss, _aux = gstep(*batch()) if feed is not None else gstep()
                    return loss
                graph_nodes = gstep.census
                mode = (f"{1 + len(gstep.graphs)} hipgraphs (fwd + decoder bwd | encoder bwd in {len(gstep.graphs)} segments), each stage's "
                        "all-reduce under the next" if gstep.graphs else "hipgraph(fwd+bwd)") + " + eager all-reduce/clip/Adam"
            except Exception as e:                       # capture is an optimisation, never a requirement
                print(f"# hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
                step, mode = eager_step, "eager"

    def barrier():
        if ddp_on:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_steps(k):
        """K steps bracketed by barrier + synchronize; -> (wall seconds of the bracket, per-step HIP-event milliseconds, last loss).  Events
        sit on the current stream, where the graph replays and the eager update are enqueued: step i = event i -> event i + 1."""
        barrier()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
        t0 = time.perf_counter()
        last = None
        for i in range(k):
            evs[i].record()
            last = step()
        evs[k].record()
        barrier()
        wall = time.perf_counter() - t0
        return wall, [evs[i].elapsed_time(evs[i + 1]) for i in range(k)], last

    graphed = "hipgraph" in mode
    for _ in range(args.warmup):
        step()
    # COLD leg: the K steps right behind the W warm-up steps (SURVEY 8d's bare protocol).  From a cold start the chip runs its first ~0.4 s
    # about 6 % slower (clock ramp from idle, tools/warm_curve.py); a training job lives in the state behind that ramp, so the headline
    # is the SETTLED leg: untimed steps for --settle-seconds, then K timed steps.  Both are in the line.
    cold_wall, cold_ms, loss = timed_steps(args.steps)
    assert torch.isfinite(loss).all(), "non-finite loss in the cold timed region"
    # Every rank runs the same NUMBER of settle steps (collectives must pair up): rank 0's clock decides.
    settle_steps = 0
    if args.settle_seconds > 0:
        t_s = time.perf_counter()
        while True:
            step()
            settle_steps += 1
            torch.cuda.synchronize()
            go_on = torch.tensor([1 if time.perf_counter() - t_s < args.settle_seconds else 0], device=dev)
            if ddp_on:
                dist.broadcast(go_on, 0)
            if not int(go_on.item()):
                break
    # events inside the timed region of an eager run must not perturb it: no gate kernels there (the intervals then include the host's launch
    # latency, as the whole eager step does); behind the region (graph mode) every timed launch sits behind a 25 us gate, see ops.KernelTimer
    timer = ops.KernelTimer(gate_us=0 if (not graphed and args.settle_seconds > 0) else 25) if (rank == 0 and not args.no_kernel_timing) else None
    if not graphed and args.settle_seconds > 0:
        ops.TIMER = timer                      # eager: per-launch HIP events inside the (settled) timed region
    if args.settle_seconds > 0:
        elapsed, step_ms, loss = timed_steps(args.steps)
    else:
        elapsed, step_ms = cold_wall, cold_ms
    ops.TIMER = None
    if not graphed and args.settle_seconds <= 0:
        graphed_like = True                    # no second leg to host the launch events: take them from steps behind the region
    else:
        graphed_like = graphed
    if graphed_like and not args.no_kernel_timing:
        # a replayed graph cannot host event records: time the same kernels (same shapes, same data) in eager steps
        # run right after the timed region, on the same stream.  Every rank runs them (their gradient all-reduces must
        # pair up across ranks); only rank 0 records events.
        ops.TIMER = timer
        for _ in range(3):
            eager_step() if args.workload == "vae" else step()
        torch.cuda.synchronize()
        ops.TIMER = None
    if ddp_on:
        dist.barrier()
    import statistics
    med, cold_med = statistics.median(step_ms), statistics.median(cold_ms)
    if ddp_on:
        tmax = torch.tensor([elapsed, med, cold_wall, cold_med], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed, med, cold_wall, cold_med = (float(v) for v in tmax.tolist())
    assert torch.isfinite(loss).all(), "non-finite loss in the timed region"

    if rank == 0:
        frames = B * T * world * args.steps
        fps = lambda ms: B * T * world / (ms * 1e-3)
        out = {
            # frames per step / MEDIAN step time by HIP events (max over ranks), settled leg; the mean over the barrier-bracketed
            # wall clock of the same K steps and the cold leg (no settle phase) are beside it
            "metric": "video frames/sec (fwd+bwd) at Bx3x16x256x256", "value": fps(med), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": med,
            "timing": {"value_from": "median of the K per-step HIP-event times, settled leg, max over ranks",
                       "settled": {"ms_per_step_median": med, "ms_per_step_wall_mean": 1e3 * elapsed / args.steps, "value_wall_mean": frames / elapsed,
                                   "ms_min": min(step_ms), "ms_max": max(step_ms)},
                       "cold": {"ms_per_step_median": cold_med, "ms_per_step_wall_mean": 1e3 * cold_wall / args.steps, "value_median": fps(cold_med),
                                "value_wall_mean": frames / cold_wall, "what": "the K steps right behind the W warm-up steps, no settle phase"}},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic" if not args.with_input_pipeline else
                    "synthetic clips on disk through the host input pipeline (worker processes -> pinned uint8 -> H2D on a side stream)",
            "rccl_ranks": world if (ddp_on and args.backend == "nccl") else 0, "grad_allreduce_dtype": args.grad_dtype if ddp_on else None,
            "config": {"workload": ((("C3" if (B, T, S) == (4, 16, 256) else "C5 per-GPU shape" if (B, T, S) == (2, 32, 256) else "custom shape")
                                     + ": full VideoVAE (enc 9 / dec 12 FactoredAttention + 3D-conv UNet) train step, ")
                                    if args.workload == "vae" else "Conv3d UNet stack alone (diagnostic), ")
                                   + f"B={B}/GPU x 3x{T}x{S}x{S}, {args.dtype} compute, fp32 params, recon+KL loss, clip+Adam",
                       "flavour": args.flavour if args.workload == "vae" else "unet", "params": nparams,
                       "global_batch": B * world, "frames_per_clip": T, "parallelism": f"dp{world}", "launch_mode": mode,
                       # node types of the captured graph(s) (hipGraphGetNodes): kernels per replayed step; memset nodes must be 0
                       "graph_nodes": graph_nodes if args.workload == "vae" else None,
                       "settle": f"{args.warmup} warm-up steps, {args.steps} timed steps (cold leg), {settle_steps} untimed steps "
                                 f"({args.settle_seconds:g} s), {args.steps} timed steps (settled leg = value)"},
        }
        summ = timer.summary() if timer is not None else {}
        nsteps_timed = 3 if graphed_like else args.steps
        timed_in = ("eager steps right after the timed region (graph replay cannot host events), every timed launch queued behind a 25 us gate kernel: "
                    "the event pair reads the kernel, not kernel + host launch latency" if graphed_like
                    else "the timed region")
        if summ:
            # the dominant hand-written kernel = the kernel (all its tagged shapes together) with the largest total time per
            # step; per-launch figures are means over its launches, which is what `rocprofv3 --stats` averages for that kernel
            # name.  Its bound is set by arithmetic intensity against the machine balance (2500 TFLOP/s / 8 TB/s = 312 FLOP/B).
            fam = {}
            for v in summ.values():
                f = fam.setdefault(v["kernel"], dict(n=0, ms=0.0, bytes=0.0, flops=0.0))
                f["n"] += v["n"]; f["ms"] += v["total_ms"]; f["bytes"] += v["bytes"] * v["n"]; f["flops"] += v["flops"] * v["n"]
            def roof(kname, top, bound=None):
                avg_ms = top["ms"] / top["n"]
                b_l, f_l = top["bytes"] / top["n"], top["flops"] / top["n"]
                gbs, tfs = b_l / (avg_ms * 1e-3) / 1e9, f_l / (avg_ms * 1e-3) / 1e12
                if bound is None:
                    bound = "mfma" if (b_l > 0 and f_l / b_l > MFMA_PEAK_TFS * 1e3 / HBM_PEAK_GBS) else "hbm"
                mf = bound == "mfma"
                return {"bound": bound, "achieved": tfs if mf else gbs, "peak": MFMA_PEAK_TFS if mf else HBM_PEAK_GBS,
                        "unit": "TFLOP/s" if mf else "GB/s", "frac": (tfs / MFMA_PEAK_TFS) if mf else (gbs / HBM_PEAK_GBS),
                        "frac_hbm": gbs / HBM_PEAK_GBS, "intensity_flop_per_byte": (f_l / b_l) if b_l > 0 else None,
                        "bound_why": f"algorithmic FLOP/B of the mean launch against the machine balance {MFMA_PEAK_TFS * 1e3 / HBM_PEAK_GBS:.0f} FLOP/B "
                                     "(2 500 TFLOP/s dense bf16 / 8 TB/s): above it the matrix roof is the lower one",
                        "traffic": measured_traffic(kname)[0], "traffic_note": measured_traffic(kname)[1],
                        "mfma_busy": measured_traffic(kname, "mfma_busy")[0], "kernel": kname, "avg_ms": avg_ms, "launches_timed": top["n"],
                        "launches_per_step": top["n"] / nsteps_timed, "alg_bytes_per_launch": b_l, "alg_flops_per_launch": f_l,
                        "alg_GBps": gbs, "alg_TFLOPps": tfs, "frac_mfma": tfs / MFMA_PEAK_TFS,
                        "kernel_ms_per_step": top["ms"] / nsteps_timed, "timed_in": timed_in}

            # `roofline`: the north-star kernel -- Conv3d forward + input gradient (conv3d_bf16_roll_kernel / conv3d_bf16_deep_kernel, every UNet layer);
            # algorithmic bytes per launch as SURVEY 8d fixes them (V*(Cin+Cout)*e).  Its bound is the lower of the two roofs for the mean launch:
            # 71 GFLOP over 152 MB = 468 FLOP/B, above the machine balance of 312, so the matrix roof (28 us per launch) lies below the HBM roof
            # (19 us) -- rounds 1-3 printed the HBM fraction under "frac"; it stays beside it as `frac_hbm`.  `roofline_step_dominant`: the
            # hand-written kernel with the largest total time per step (all its shapes together), bound by the same rule.
            conv_k = next((k for k in fam if k.startswith("conv3d_bf16")), None) or next((k for k in fam if k.startswith("conv3d")), None)
            kname, top = max(fam.items(), key=lambda kv: kv[1]["ms"])
            if conv_k is not None:
                out["roofline"] = roof(conv_k, fam[conv_k])
                out["roofline_step_dominant"] = roof(kname, top)
            else:
                out["roofline"] = roof(kname, top)
            rows = sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"])[:12]
            out["kernels"] = [{"launch": k, "kernel": v["kernel"], "per_step": v["n"] / nsteps_timed, "avg_ms": round(v["avg_ms"], 4),
                               "ms_per_step": round(v["total_ms"] / nsteps_timed, 3),
                               "GBps": round(v["bytes"] / v["avg_ms"] / 1e6, 1), "TFLOPps": round(v["flops"] / v["avg_ms"] / 1e9, 1)}
                              for k, v in rows]
        conv = {k: v for k, v in summ.items() if k.startswith(("conv3d", "convt"))}        # every conv-like launch of the UNet
        if conv:
            tot_ms = sum(v["total_ms"] for v in conv.values()) / nsteps_timed
            tot_b = sum(v["bytes"] * v["n"] for v in conv.values()) / nsteps_timed
            tot_f = sum(v["flops"] * v["n"] for v in conv.values()) / nsteps_timed
            out["conv_stack"] = {"ms_per_step": tot_ms, "alg_GB_per_step": tot_b / 1e9, "GBps": tot_b / tot_ms / 1e6,
                                 "frac_hbm": tot_b / tot_ms / 1e6 / HBM_PEAK_GBS, "tflops": tot_f / tot_ms / 1e9,
                                 "frames_per_s_conv_only": B * T / (tot_ms * 1e-3)}
            if os.environ.get("VVAE_BENCH_VERBOSE"):
                for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"]):
                    print(f"# {k:44s} n={v['n']:3d} avg {v['avg_ms']:8.3f} ms  {v['bytes'] / v['avg_ms'] / 1e6:8.1f} GB/s "
                          f"{v['flops'] / v['avg_ms'] / 1e9:8.1f} TF/s", file=sys.stderr)
        if world == 1 and not ddp_on and not args.no_also and args.workload == "vae" and graphed and (args.flavour, B, T, S) == ("model", 4, 16, 256):
            # free the headline's model first: the secondary runs start from an empty card and after everything the headline measures
            import gc
            step = eager_step = gstep = model = opt = None
            gc.collect()
            torch.cuda.empty_cache()
            out["also"] = also_runs(args, dev, dtype)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out))
    if ddp_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
