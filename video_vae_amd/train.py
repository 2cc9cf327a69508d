#!/usr/bin/env python3
"""Training driver: counterpart of the reference's train/rl_nonadversarial.py `__main__` (:216-391) and of the multi-host
loop claude_distributed/distributed_train.py (:433-583), for one node of MI355X GPUs.

    python -m video_vae_amd.train --steps 100                               # 1 GPU, synthetic clips
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m video_vae_amd.train --steps 100

Same constants (rl_nonadversarial.py:36-57), model config (:234-236), optimizer (:241-253), hparams (:255-263), batch/frames
curriculum (:287-295) and log keys (:344-359).  Data: any iterable of {"video": float32 (B,T,H,W,3) in [0,1], "mask": float32
(B,T)} batches (the reference's dataloader contract, train/dataloader.py:387-390): ``--data DIR`` streams clips from DIR through
video_vae_amd/data.py (worker processes -> pinned uint8 batches -> H2D on a side stream, per-rank shuffle seed + rank); without
--data this driver feeds seeded synthetic clips.  One process per GPU; gradients are all-reduced over RCCL overlapped with backward (ddp.py); rank 0 logs;
SIGTERM/SIGINT flips a flag and the loop checkpoints and exits (distributed_train.py:58-67,489-494).

The step runs on the path bench.py measures: ``StepRunner`` keeps one captured ``GraphedTrainStep`` per (batch, frames) shape of the
curriculum -- the counterpart of the reference's ``nnx.jit(train_step)`` (rl_nonadversarial.py:276-277,332), which also compiles once
per shape -- and runs eagerly only until a shape has been seen ``--capture_after`` times (a shape met once is not worth a capture) or
with ``--eager``.  Every ``--sample_every`` steps the reconstruction and the original of one clip are written with data.batch_to_video
(rl_nonadversarial.py:337-343); after every epoch ``--eval_steps`` batches go through eval_step (:200-208,362-391).
"""
import argparse
import gc
import math
import os
import signal
import statistics
import time

import torch
import torch.distributed as dist

import video_vae_amd as V
from video_vae_amd import ddp, loss as L, optim, rl_model

NUM_EPOCHS, BATCH_SIZE, MAX_FRAMES, RESIZE, SEED = 100, 2, 32, (256, 256), 0
NEGATIVE_PENALTY_TRAINING_STEPS = 2000
_SHOULD_STOP = False
TIMES = []                    # (start event, end event, launch mode) per step, read after the loop


def _stop(signum, frame):
    global _SHOULD_STOP
    _SHOULD_STOP = True


def synthetic_batches(batch, frames, size, seed, steps, device, pool=4):
    """Seeded synthetic clips with ragged lengths: a pool of ``pool`` device-resident batches, cycled (generating 12.6 M uniform floats
    per step on the host costs more than the train step it would feed)."""
    g = torch.Generator().manual_seed(seed)
    made = []
    for _ in range(min(pool, steps)):
        video = torch.rand((batch, frames, size[0], size[1], 3), generator=g)
        lens = torch.randint(max(1, frames // 2), frames + 1, (batch,), generator=g)
        mask = (torch.arange(frames)[None, :] < lens[:, None]).float()
        made.append({"video": video.to(device), "mask": mask.to(device)})
    for i in range(steps):
        yield made[i % len(made)]


class StepRunner:
    """train_step(video, mask, hparams) -> (loss, aux), replayed from a captured hipGraph once a shape has been seen often enough.

    One graph per input shape; a change of ``hparams`` (the compression-rate switch at step 2000, :327-328) re-captures that shape's
    graph, since the loss constants are baked into the captured launches.  Each graph draws its noise from its own ``Rngs`` (a captured
    step pins the draws of ITS rngs to static buffers; the eager fallback keeps drawing from the driver's)."""

    def __init__(self, model, opt, hw, rngs, perceptual_loss_fn=None, vgg_params=None, use_graph=True, capture_after=1, log=None, enc_segments=3):
        self.model, self.opt, self.hw, self.rngs = model, opt, hw, rngs
        self.ploss, self.vgg_params = perceptual_loss_fn, vgg_params
        self.use_graph, self.capture_after, self.log = use_graph, capture_after, log or (lambda msg: None)
        self.graphs, self.seen = {}, {}
        self.enc_segments = enc_segments
        self.mode = "eager"

    def __call__(self, video, mask, hparams):
        shape = (tuple(video.shape), video.dtype)
        hkey = tuple(sorted(hparams.items()))
        hit = self.graphs.get(shape)
        if hit is not None and hit[0] == hkey:
            self.mode = "hipgraph"
            loss, aux = hit[1](video, mask)
            return loss, dict(aux)
        n = self.seen[shape] = self.seen.get(shape, 0) + 1
        if self.use_graph and n > self.capture_after:
            from .graph import GraphedTrainStep
            if hit is not None:                                  # same shape, new loss constants: drop the old graph and its pool first
                del self.graphs[shape], hit
                gc.collect()
                torch.cuda.empty_cache()
            try:
                t0 = time.perf_counter()
                gc.collect()                                     # no autograd graph of an eager step may outlive into the capture
                g = GraphedTrainStep(self.model, self.opt, video, mask, dict(hparams), self.hw, V.Rngs(1_000_003 * (self.rngs.seed + 1) + len(self.seen)),
                                     perceptual_loss_fn=self.ploss, vgg_params=self.vgg_params, stream=torch.cuda.current_stream(),
                                     enc_segments=self.enc_segments)
                self.graphs[shape] = (hkey, g)
                self.log(f"captured the train step for video {tuple(video.shape)} in {time.perf_counter() - t0:.1f} s "
                         f"({1 + len(g.graphs)} hipGraph{'s' if g.graphs else ''})")
                self.mode = "hipgraph"
                loss, aux = g(video, mask)
                return loss, dict(aux)
            except Exception as e:                               # capture is an optimisation, never a requirement
                self.log(f"hipGraph capture failed ({type(e).__name__}: {e}); this shape stays eager")
                self.seen[shape] = -(1 << 30)
        self.mode = "eager"
        loss, aux = L.train_step(self.model, self.opt, video, mask, hparams, self.hw, self.rngs, self.ploss, self.vgg_params)
        return loss, aux


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20, help="steps per epoch for synthetic data")
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--per_device_batch_size", type=int, default=BATCH_SIZE)
    ap.add_argument("--max_frames", type=int, default=MAX_FRAMES)
    ap.add_argument("--size", type=int, default=RESIZE[0])
    ap.add_argument("--flavour", default="rl", choices=["rl", "model"])
    ap.add_argument("--model_path", type=str, default=None, help="checkpoint directory to resume from")
    ap.add_argument("--save_dir", type=str, default=None)
    ap.add_argument("--small", action="store_true", help="tiny model (depth 1) for smoke runs")
    ap.add_argument("--data", type=str, default=None, help="directory of clips (videos{i}/*.npy|npz|mp4...): the host input pipeline")
    ap.add_argument("--num_workers", type=int, default=4)
    ap.add_argument("--grad-dtype", default="f32", choices=["f32", "bf16"], help="dtype of the gradient all-reduce (ddp.GradReducer)")
    ap.add_argument("--enc-segments", type=int, default=3, help="data parallel: hipGraphs the encoder's backward is cut into (graph.py)")
    ap.add_argument("--eager", action="store_true", help="never capture: every step through the eager L.train_step")
    ap.add_argument("--capture_after", type=int, default=1, help="eager steps of a (batch, frames) shape before its step is captured as a hipGraph")
    ap.add_argument("--log_every", type=int, default=10)
    ap.add_argument("--sample_every", type=int, default=0,
                    help="every N steps write the reconstruction and the original of one clip under --sample_dir (rl_nonadversarial.py:337-343: 500)")
    ap.add_argument("--sample_dir", type=str, default=None)
    ap.add_argument("--sample_ext", default="npz", choices=["npz", "npy", "mp4"], help="mp4 needs ffmpeg on PATH (data.batch_to_video)")
    ap.add_argument("--eval_steps", type=int, default=0, help="batches per epoch through eval_step after the epoch's training (:362-391)")
    ap.add_argument("--eval_data", type=str, default=None, help="directory of evaluation clips (default: synthetic / --data)")
    ap.add_argument("--vgg", type=str, default=None,
                    help="perceptual loss (rl flavour; rl_nonadversarial.py:125,272-274): 'random' = randomly initialised VGG16 head, or the "
                         "path of an .npz / .pt with its six tensors (the ImageNet weights are a remote download in the reference)")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC for RCCL between the ranks; before the first GPU call
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # ONE non-default stream for the whole run: model construction, eager steps, graph captures and replays, the optimizer update and
    # (through the current-stream edges of ProcessGroupNCCL) the collectives.  Autograd pins every parameter's AccumulateGrad node to the
    # stream that was current when the node was made; a capture on another stream than the eager steps before it crashes the engine.
    train_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(train_stream)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    signal.signal(signal.SIGTERM, _stop)
    signal.signal(signal.SIGINT, _stop)

    size, patch = args.size, 16
    hw = (size // patch) ** 2
    cfg = dict(height=size, width=size, channels=3, patch_size=patch, encoder_depth=9, decoder_depth=12, mlp_dim=1536,
               num_heads=8, qkv_features=512, max_temporal_len=64, spatial_compression_rate=8, unembedding_upsample_rate=4)
    if args.small:
        cfg.update(encoder_depth=1, decoder_depth=1, mlp_dim=256, qkv_features=128, num_heads=4)
    cls = rl_model.VideoVAE if args.flavour == "rl" else V.VideoVAE
    model = cls(rngs=V.Rngs(2), **cfg).to(dev)
    opt = optim.Optimizer(model, optim.reference_schedule(batch_size=args.per_device_batch_size * world))
    red = ddp.GradReducer(opt, grad_dtype=torch.bfloat16 if args.grad_dtype == "bf16" else torch.float32) if world > 1 else None
    hparams = dict(L.HPARAMS)
    if args.model_path:
        if rank == 0 or world == 1:
            V.load_checkpoint(model, opt, args.model_path)
        hparams["max_compression_rate"] = 100000                  # rl_nonadversarial.py:265-268
    if red is not None:
        # replicate rank 0's parameters, Adam moments and update count (start-up and resume: rank-0 restore + broadcast of
        # {"model", "optimizer"}, claude_distributed/distributed_train.py:321-341,378-380)
        red.broadcast_state(0)
    if rank == 0:
        print(f"Trainable Parameters: {sum(p.numel() for p in model.parameters()) / 1e6} Million", flush=True)

    ploss = vgg_params = None
    if args.vgg:
        from video_vae_amd import perceptual
        vgg, vgg_params = perceptual.load_vgg(None if args.vgg == "random" else args.vgg, device=dev)
        ploss = perceptual.get_adversarial_perceptual_loss_fn(vgg)
    rngs = V.Rngs(3 + rank)
    log = (lambda msg: print(msg, flush=True)) if rank == 0 else (lambda msg: None)
    runner = StepRunner(model, opt, hw, rngs, ploss, vgg_params, use_graph=not args.eager, capture_after=args.capture_after, log=log,
                        enc_segments=args.enc_segments)
    if args.sample_every and not args.sample_dir:
        ap.error("--sample_every needs --sample_dir")

    def loader(epoch, bsz, frames, directory, salt):
        if directory:
            from video_vae_amd import data as D
            host = D.create_batched_dataloader(directory, batch_size=bsz, max_frames=frames, resize=(size, size), crop_size=size,
                                               shuffle=True, seed=SEED + epoch, num_workers=args.num_workers, prefetch_size=16,
                                               drop_remainder=True, rank=rank, num_epochs=1, as_uint8=True)
            return D.DevicePrefetcher(host, dev, dtype=torch.bfloat16)       # the cast of :330 rides in the H2D side stream
        return synthetic_batches(bsz, frames, (size, size), SEED + epoch + 1000 * rank + salt, max(args.steps, args.eval_steps), dev)

    def dump(tag, epoch, i, batch, recon, bsz):
        from video_vae_amd import data as D
        d = os.path.join(args.sample_dir, f"{tag}/epoch{epoch}")
        os.makedirs(d, exist_ok=True)
        D.batch_to_video({"video": recon[:bsz], "mask": batch["mask"]}, os.path.join(d, f"video_{i}_latent.{args.sample_ext}"), fps=30.0)
        D.batch_to_video(batch, os.path.join(d, f"video_{i}_original.{args.sample_ext}"), fps=30.0)

    start, global_step = time.perf_counter(), 0
    for epoch in range(args.epochs):
        max_mult = min(int(math.log2(max(args.per_device_batch_size, 1))), int(math.log2(64 / args.max_frames)) - 1)
        mult = max(0, min(epoch, max_mult))                       # batch <-> frames curriculum, :287-295
        bsz, frames = args.per_device_batch_size // (2 ** mult), args.max_frames * (2 ** mult)
        for i, batch in enumerate(loader(epoch, bsz, frames, args.data, 0)):
            if _SHOULD_STOP or i >= args.steps:
                break
            if i > NEGATIVE_PENALTY_TRAINING_STEPS:
                hparams["max_compression_rate"] = 10000
            video = batch["video"].to(torch.bfloat16)             # :330
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()                                           # step i = this event -> the next step's (bench.py's per-step clock)
            loss, aux = runner(video, batch["mask"], hparams)
            TIMES.append((ev, runner.mode))
            global_step += 1
            if args.sample_every and i % args.sample_every == args.sample_every - 1 and rank == 0:
                recon = aux.get("reconstruction")
                if recon is None:                                 # a replayed step keeps no reconstruction: one eval pass on this batch
                    recon = L.eval_step(model, video, batch["mask"], hparams, hw, rngs, ploss, vgg_params)[1]["reconstruction"]
                dump("train", epoch, i, batch, recon, bsz)
            if i % args.log_every == 0 or i == args.steps - 1:
                keys = [k for k in aux if k != "reconstruction"]
                vals = [loss] + [aux[k] for k in keys]
                if world > 1:
                    vals = ddp.all_reduce_mean_scalars(vals)
                if rank == 0:
                    msg = ", ".join(f"{k} = {float(v):.4f}" for k, v in zip(["Loss"] + keys, vals))
                    print(f"Epoch {epoch}, Step {i}: {msg}, lr = {opt.last_lr:.3e}, time = {time.perf_counter() - start:.2f}, "
                          f"mode = {runner.mode}, effective_batch_size = {bsz}, effective_max_frames = {frames}", flush=True)
        if args.save_dir and rank == 0:
            tag = "checkpoint_sigterm" if _SHOULD_STOP else "checkpoint"
            V.save_checkpoint(model, opt, os.path.join(args.save_dir, f"{tag}_{epoch}"))
        if args.eval_steps and not _SHOULD_STOP:
            for i, batch in enumerate(loader(epoch, bsz, frames, args.eval_data or args.data, 500_000)):
                if _SHOULD_STOP or i >= args.eval_steps:
                    break
                video = batch["video"].to(torch.bfloat16)
                loss, aux = L.eval_step(model, video, batch["mask"], hparams, hw, rngs, ploss, vgg_params)
                if args.sample_every and i % 100 == 0 and rank == 0:
                    dump("eval", epoch, i, batch, aux["reconstruction"], bsz)
                keys = [k for k in aux if k != "reconstruction"]
                vals = [loss] + [aux[k] for k in keys]
                if world > 1:
                    vals = ddp.all_reduce_mean_scalars(vals)
                if rank == 0:
                    msg = ", ".join(f"{k} = {float(v):.4f}" for k, v in zip(["Loss"] + keys, vals))
                    print(f"VALIDATION Epoch {epoch}, Step {i}: {msg}, effective_batch_size = {bsz}, effective_max_frames = {frames}", flush=True)
        if world > 1:
            dist.barrier()
        if _SHOULD_STOP:
            break
    end = torch.cuda.Event(enable_timing=True)
    end.record()
    torch.cuda.synchronize()
    if rank == 0 and TIMES:
        by_mode = {}
        for (a, mode), (b, mode_next) in zip(TIMES, TIMES[1:] + [(end, None)]):
            if mode_next in (mode, None):                              # a step followed by a capture is not a steady-state step
                by_mode.setdefault(mode, []).append(a.elapsed_time(b))
        for mode, ts in by_mode.items():
            tail = ts[len(ts) // 5:] if len(ts) >= 10 else ts          # the first fifth carries clock ramp and allocator warm-up
            print(f"train summary: {len(ts)} {mode} steps, median {statistics.median(tail):.2f} ms/step over the last {len(tail)} "
                  f"(logging every {args.log_every} steps included)", flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
