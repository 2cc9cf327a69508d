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
"""
import argparse
import math
import os
import signal
import time

import torch
import torch.distributed as dist

import video_vae_amd as V
from video_vae_amd import ddp, loss as L, optim, rl_model

NUM_EPOCHS, BATCH_SIZE, MAX_FRAMES, RESIZE, SEED = 100, 2, 32, (256, 256), 0
NEGATIVE_PENALTY_TRAINING_STEPS = 2000
_SHOULD_STOP = False


def _stop(signum, frame):
    global _SHOULD_STOP
    _SHOULD_STOP = True


def synthetic_batches(batch, frames, size, seed, steps, device):
    g = torch.Generator().manual_seed(seed)
    for _ in range(steps):
        video = torch.rand((batch, frames, size[0], size[1], 3), generator=g)
        lens = torch.randint(max(1, frames // 2), frames + 1, (batch,), generator=g)
        mask = (torch.arange(frames)[None, :] < lens[:, None]).float()
        yield {"video": video.to(device), "mask": mask.to(device)}


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
    red = ddp.GradReducer(opt) if world > 1 else None
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
    start, global_step = time.perf_counter(), 0
    for epoch in range(args.epochs):
        max_mult = min(int(math.log2(max(args.per_device_batch_size, 1))), int(math.log2(64 / args.max_frames)) - 1)
        mult = max(0, min(epoch, max_mult))                       # batch <-> frames curriculum, :287-295
        bsz, frames = args.per_device_batch_size // (2 ** mult), args.max_frames * (2 ** mult)
        if args.data:
            from video_vae_amd import data as D
            host = D.create_batched_dataloader(args.data, batch_size=bsz, max_frames=frames, resize=(size, size), crop_size=size,
                                               shuffle=True, seed=SEED + epoch, num_workers=args.num_workers, prefetch_size=16,
                                               drop_remainder=True, rank=rank, num_epochs=1, as_uint8=True)
            batches = D.DevicePrefetcher(host, dev, dtype=torch.float32)
        else:
            batches = synthetic_batches(bsz, frames, (size, size), SEED + epoch + 1000 * rank, args.steps, dev)
        for i, batch in enumerate(batches):
            if _SHOULD_STOP or i >= args.steps:
                break
            if i > NEGATIVE_PENALTY_TRAINING_STEPS:
                hparams["max_compression_rate"] = 10000
            video = batch["video"].to(torch.bfloat16)             # :330
            loss, aux = L.train_step(model, opt, video, batch["mask"], hparams, hw, rngs, ploss, vgg_params)
            global_step += 1
            if i % 10 == 0 or i == args.steps - 1:
                keys = [k for k in aux if k != "reconstruction"]
                vals = [loss] + [aux[k] for k in keys]
                if world > 1:
                    vals = ddp.all_reduce_mean_scalars(vals)
                if rank == 0:
                    msg = ", ".join(f"{k} = {float(v):.4f}" for k, v in zip(["Loss"] + keys, vals))
                    print(f"Epoch {epoch}, Step {i}: {msg}, lr = {opt.last_lr:.3e}, time = {time.perf_counter() - start:.2f}", flush=True)
        if args.save_dir and rank == 0:
            tag = "checkpoint_sigterm" if _SHOULD_STOP else "checkpoint"
            V.save_checkpoint(model, opt, os.path.join(args.save_dir, f"{tag}_{epoch}"))
        if world > 1:
            dist.barrier()
        if _SHOULD_STOP:
            break
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
