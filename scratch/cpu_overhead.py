import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
for wl in ("unet", "vae"):
    sys.argv = ["bench.py", "--workload", wl]
    args = bench.parse()
    import video_vae_amd as V
    from video_vae_amd import ops, optim, loss as L
    dev = torch.device("cuda", 0)
    model, cfg = bench.build_model(args, dev, torch.bfloat16)
    opt = optim.Optimizer(model, 1e-5)
    g = torch.Generator().manual_seed(0)
    B, T, S = 4, 16, 256
    if wl == "unet":
        feat = (torch.randn((B, T, S, S, 12), generator=g) * 0.5).to(dev, torch.bfloat16)
        tgt = torch.rand((B, T, S, S, 3), generator=g).to(dev, torch.bfloat16)
        mask = torch.ones((B, T), device=dev)
        def step():
            opt.zero_grad(); recon = model(feat); mse, _ = ops.masked_mse_mae(tgt, recon, mask, 1); mse.mean().backward(); opt.update()
    else:
        video = torch.rand((B, T, S, S, 3), generator=g).to(dev, torch.bfloat16); mask = torch.ones((B, T), device=dev); rngs = V.Rngs(3)
        def step():
            L.train_step(model, opt, video, mask, L.HPARAMS, 256, rngs)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(wl, "cpu enqueue ms", (t1 - t0) * 1e3, "gpu tail ms", (t2 - t1) * 1e3, "total", (t2 - t0) * 1e3)
    del model, opt
    torch.cuda.empty_cache()
