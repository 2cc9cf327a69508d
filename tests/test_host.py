"""CPU: host logic of the product package -- C-ABI surface, Rngs stand-in, schedule, gradient bucketing, and the
world_size=2 gloo rehearsal of the data-parallel reducer.  No HIP compute is called here (there is no GPU)."""
import ctypes
import json
import math
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    from video_vae_amd._lib import parse_header, LIB_PATH, lib
    protos = parse_header()
    assert len(protos) >= 35
    assert os.path.exists(LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    raw = ctypes.CDLL(LIB_PATH)
    for name in protos:
        assert hasattr(raw, name), f"{name} declared in include/vvae_hip.h but not exported"
    out = subprocess.run(["nm", "-D", "--defined-only", LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln and "vvae_" in ln}
    assert exported == set(protos), (exported ^ set(protos))
    lib()      # argtypes bind


def test_ops_fail_loudly_without_gpu():
    """No CPU fallback: a CPU tensor must raise, not silently compute."""
    import video_vae_amd as V
    from video_vae_amd._lib import VvaeError
    m = V.UNet(4, 8, 1, 3, V.Rngs(0), dtype=torch.float32)
    with pytest.raises(VvaeError):
        m(torch.zeros(1, 2, 8, 8, 4))
    with pytest.raises(VvaeError):
        V.ops.group_norm_silu(torch.zeros(1, 2, 4, 4, 8), torch.ones(8), torch.zeros(8), 8)


def test_product_never_imports_oracle():
    code = "import sys; import video_vae_amd; import bench; sys.exit(1 if any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules) else 0)"
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-400:]


def test_surface_matches_reference_signatures():
    """constructor / call surface of the reference classes."""
    import inspect
    import video_vae_amd as V
    from video_vae_amd import rl_model
    assert list(inspect.signature(V.UNet.__init__).parameters)[1:] == [
        "channels", "base_features", "num_levels", "out_features", "rngs", "temporal_kernel", "dtype", "param_dtype"]
    vae = ["height", "width", "channels", "patch_size", "encoder_depth", "decoder_depth", "mlp_dim", "num_heads", "qkv_features",
           "max_temporal_len", "spatial_compression_rate", "unembedding_upsample_rate", "rngs", "dtype", "param_dtype"]
    assert list(inspect.signature(V.VideoVAE.__init__).parameters)[1:] == vae
    assert list(inspect.signature(rl_model.VideoVAE.__init__).parameters)[1:] == vae
    assert list(inspect.signature(V.VideoVAE.forward).parameters)[1:] == ["x", "mask", "rngs", "train"]
    assert list(inspect.signature(V.load_checkpoint).parameters) == ["model", "optimizer", "path"]
    assert list(inspect.signature(V.save_checkpoint).parameters) == ["model", "optimizer", "path"]
    assert list(inspect.signature(V.FactoredAttention.__init__).parameters)[1:8] == [
        "mlp_dim", "in_features", "num_heads", "qkv_features", "max_temporal_len", "max_spatial_len", "rngs"]
    m = V.VideoVAE(32, 32, 3, 8, 1, 1, 64, 4, 32, 8, 4, 4, V.Rngs(2))
    assert hasattr(m, "encoder") and hasattr(m, "decoder") and m.fill_token.shape == (1, 1, 1, 48)
    assert sum(p.numel() for p in V.UNet(12, 16, 3, 3, V.Rngs(0)).parameters()) == 1_384_927
    assert float(m.decoder.unet.final_conv.kernel.abs().max()) == 0.0                          # zero-init, unet.py:150


def test_rngs_streams_and_injection():
    import video_vae_amd as V
    a, b = V.Rngs(3), V.Rngs(3)
    x1 = a.draw("reparam_eps", "normal", (4, 5), "cpu")
    x2 = b.draw("reparam_eps", "normal", (4, 5), "cpu")
    assert torch.equal(x1, x2)
    assert not torch.equal(a.draw("reparam_eps", "normal", (4, 5), "cpu"), x1)          # a fresh key per call
    assert not torch.equal(V.Rngs(4).draw("reparam_eps", "normal", (4, 5), "cpu"), x1)  # seed matters
    inj = torch.ones(4, 5)
    a.inject("reparam_eps", inj)
    assert torch.equal(a.draw("reparam_eps", "normal", (4, 5), "cpu"), inj)
    u = V.Rngs(0).draw("u", "uniform", (1000,), "cpu")
    assert 0 <= float(u.min()) and float(u.max()) < 1


def test_schedule_matches_reference_constants():
    from video_vae_amd import optim
    s = optim.reference_schedule(batch_size=2)
    assert s(0) == 0.0
    assert abs(s(14142) - 2e-5) < 1e-10
    assert abs(s(7071) - 1e-5) < 1e-9
    assert abs(s(10_000_000) - 2e-6) < 1e-12
    mid = 14142 + (1_000_000 - 14142) / 2
    assert abs(s(mid) - (2e-6 + (2e-5 - 2e-6) * 0.5)) < 1e-9


def _toy():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Tanh(),
                               torch.nn.Linear(16, 3))


def test_gradient_landing_in_flat_buffer():
    """Bucketed landing: grads end up in the flat buffer in reverse registration order; unused params land zeros."""
    from video_vae_amd import optim
    m = _toy()
    extra = torch.nn.Parameter(torch.ones(5))
    m.register_parameter("unused", extra)
    opt = optim.Optimizer(m, 1e-3, bucket_bytes=256)
    assert opt.names[0] == "4.bias" and opt.names[-1] == "unused"          # reverse registration order
    assert len(opt.buckets) > 2 and opt.buckets[0][0] == 0 and opt.buckets[-1][1] == opt.numel
    x = torch.randn(9, 6)
    ref = _toy()
    ref(x).square().mean().backward()
    opt.zero_grad()
    m(x).square().mean().backward()
    for b in range(len(opt.buckets)):
        if not opt.landed[b]:
            opt._land(b)
    for n, p, gv in zip(opt.names, opt.params, opt.gviews):
        if n == "unused":
            assert float(gv.abs().max()) == 0.0
        else:
            assert torch.allclose(gv, dict(ref.named_parameters())[n].grad, atol=1e-7), n
        assert p.grad is None
    assert all(p.data_ptr() >= opt.p.data_ptr() for p in opt.params)         # parameters alias the flat buffer


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _ddp_worker(rank, world, port, out, grad_dtype="float32"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        from video_vae_amd import optim, ddp
        torch.set_num_threads(1)
        m = _toy()
        if rank == 1:                                   # replicas start different: broadcast must repair it
            with torch.no_grad():
                for p in m.parameters():
                    p.add_(1.0)
        opt = optim.Optimizer(m, 1e-2, bucket_bytes=256, bf16_shadow=False)
        red = ddp.GradReducer(opt, grad_dtype=getattr(torch, grad_dtype))
        red.broadcast_parameters(0)
        g = torch.Generator().manual_seed(100 + rank)   # per-rank shard: seed + rank
        res = {}
        for step in range(3):
            x = torch.randn((5, 6), generator=g)
            opt.zero_grad()
            m(x).square().mean().backward()
            for b in range(len(opt.buckets)):
                if not opt.landed[b]:
                    opt._land(b)
            red.finish()
            assert len(red.handles) == 0
            res[f"g{step}"] = opt.g.clone()
            res[f"x{step}"] = x
            with torch.no_grad():                        # stand-in for the HIP Adam kernel: same update on every rank
                opt.p.add_(opt.g, alpha=-0.05 / world)
            res[f"p{step}"] = opt.p.clone()
        (mean_loss,) = ddp.all_reduce_mean_scalars([torch.tensor(float(rank))])
        res["mean"] = mean_loss
        torch.save(res, os.path.join(out, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_ddp_gloo_world2(tmp_path):
    """Replicas: distinct shard per rank, bucketed SUM all-reduce, bit-identical parameters after every update
    (reference properties claude_distributed/test_distributed.py:75-97,159-163)."""
    port = _free_port()
    mp.spawn(_ddp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    assert float(r0["mean"]) == 0.5 and float(r1["mean"]) == 0.5
    from video_vae_amd import optim
    m = _toy()
    opt = optim.Optimizer(m, 1e-2, bucket_bytes=256, bf16_shadow=False)
    for step in range(3):
        assert not torch.equal(r0[f"x{step}"], r1[f"x{step}"])                  # shards differ
        assert torch.equal(r0[f"g{step}"], r1[f"g{step}"])                      # all-reduced gradients identical
        assert torch.equal(r0[f"p{step}"], r1[f"p{step}"])                      # replicas stay bit-identical
        want = torch.zeros_like(opt.g)
        for x in (r0[f"x{step}"], r1[f"x{step}"]):                              # = sum of the per-rank gradients
            opt.zero_grad()
            m(x).square().mean().backward()
            for b in range(len(opt.buckets)):
                if not opt.landed[b]:
                    opt._land(b)
            want += opt.g
        assert torch.allclose(r0[f"g{step}"], want, rtol=1e-5, atol=1e-7)
        with torch.no_grad():
            opt.p.add_(want, alpha=-0.05 / 2)


def test_ddp_gloo_world2_bf16_gradient_allreduce(tmp_path):
    """--grad-dtype bf16: buckets cross the transport in bf16 (half the bytes over xGMI).  Replicas still hold bit-identical gradients and
    parameters after every update; against the fp32 reduction (the reference's arithmetic) the all-reduced gradient is within bf16
    rounding of the sum of the shard gradients and the 3-step parameter trajectory within that rounding times the step size."""
    port = _free_port()
    mp.spawn(_ddp_worker, args=(2, port, str(tmp_path), "bfloat16"), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    (tmp_path / "f32").mkdir()
    mp.spawn(_ddp_worker, args=(2, _free_port(), str(tmp_path / "f32")), nprocs=2, join=True)
    f0 = torch.load(tmp_path / "f32" / "r0.pt")
    for step in range(3):
        assert torch.equal(r0[f"g{step}"], r1[f"g{step}"]) and torch.equal(r0[f"p{step}"], r1[f"p{step}"])      # replicas identical
        g, ref = r0[f"g{step}"], f0[f"g{step}"]
        assert not torch.equal(g, ref), "the bf16 path did not run"
        assert torch.equal(g, g.to(torch.bfloat16).float()), "all-reduced values are bf16 numbers"
        # two bf16-rounded addends (2^-9 relative each) and one bf16 sum (2^-9 of the result), measured against the gradient's scale:
        # (the trajectories differ by then, so the bound is on the first step; later steps only drift by lr x that)
        if step == 0:
            assert float((g - ref).abs().max()) <= 3 * 2.0 ** -9 * float(ref.abs().max())
        assert float((r0[f"p{step}"] - f0[f"p{step}"]).abs().max()) <= (step + 1) * 0.05 * 3 * 2.0 ** -8 * float(f0["g0"].abs().max()) + 1e-6


def _adam_standin(opt, lr_fn):
    """What the fused HIP clip+Adam kernel does to (p, m, v, count), in torch ops (CPU stand-in for the resume test)."""
    lr = lr_fn(opt.count)
    opt.count += 1
    with torch.no_grad():
        opt.m.mul_(0.9).add_(opt.g, alpha=0.1)
        opt.v.mul_(0.999).addcmul_(opt.g, opt.g, value=0.001)
        c1, c2 = 1 - 0.9 ** opt.count, 1 - 0.999 ** opt.count
        opt.p.sub_(lr * (opt.m / c1) / ((opt.v / c2).sqrt() + 1e-8))


def _resume_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        import video_vae_amd as V
        from video_vae_amd import optim, ddp
        torch.set_num_threads(1)
        sched = optim.warmup_cosine_decay_schedule(0.0, 1e-2, 10, 100, 1e-3)
        ck = os.path.join(out, "ck")
        if rank == 0:                                   # a run that took 3 updates and checkpointed
            m0 = _toy()
            o0 = optim.Optimizer(m0, sched, bucket_bytes=256, bf16_shadow=False)
            g = torch.Generator().manual_seed(7)
            for _ in range(3):
                o0.zero_grad()
                m0(torch.randn((5, 6), generator=g)).square().mean().backward()
                for b in range(len(o0.buckets)):
                    if not o0.landed[b]:
                        o0._land(b)
                _adam_standin(o0, sched)
            V.save_checkpoint(m0, o0, ck)
        dist.barrier()
        # resume as video_vae_amd/train.py does: fresh model on every rank, rank 0 restores, state is broadcast
        torch.manual_seed(100 + rank)
        m = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3))
        opt = optim.Optimizer(m, sched, bucket_bytes=256, bf16_shadow=False)
        red = ddp.GradReducer(opt)
        if rank == 0:
            V.load_checkpoint(m, opt, ck)
        red.broadcast_state(0)
        res = {"count0": opt.count, "p0": opt.p.clone(), "m0": opt.m.clone(), "v0": opt.v.clone()}
        x = torch.randn((5, 6), generator=torch.Generator().manual_seed(200 + rank))
        opt.zero_grad()
        m(x).square().mean().backward()
        for b in range(len(opt.buckets)):
            if not opt.landed[b]:
                opt._land(b)
        red.finish()
        with torch.no_grad():
            opt.g.div_(world)
        _adam_standin(opt, sched)
        res.update(count1=opt.count, p1=opt.p.clone(), m1=opt.m.clone(), v1=opt.v.clone())
        torch.save(res, os.path.join(out, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_ddp_resume_broadcasts_optimizer_state(tmp_path):
    """Multi-rank resume: rank 0 restores the checkpoint, parameters AND Adam moments AND the update count are broadcast
    (reference claude_distributed/distributed_train.py:321-341 broadcasts {"model", "optimizer"}); after the next update the
    replicas are bit-identical.  With parameters only, rank 1 would restart at count 0 (lr = schedule(0) = 0, zero moments)."""
    port = _free_port()
    mp.spawn(_resume_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    assert r0["count0"] == 3 and r1["count0"] == 3
    assert float(r0["m0"].abs().max()) > 0 and float(r0["v0"].abs().max()) > 0
    for k in ("p0", "m0", "v0", "p1", "m1", "v1"):
        assert torch.equal(r0[k], r1[k]), k
    assert r0["count1"] == 4 and r1["count1"] == 4
    assert not torch.equal(r0["p0"], r0["p1"])


def test_bench_self_launch_relays_rank0_line(tmp_path):
    """`bench.py --gpus N` started bare (no WORLD_SIZE) must start the ranks itself: here the launcher half is exercised with a
    stand-in rank script (no GPU): the parent relays the one JSON line and propagates a failing rank's exit code."""
    import importlib.util
    import subprocess
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    fake = tmp_path / "fake_rank.py"
    fake.write_text(
        "import os, sys, json\n"
        "r = int(os.environ['RANK']); w = int(os.environ['WORLD_SIZE'])\n"
        "assert os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
        "if '--fail' in sys.argv and r == 1: sys.exit(3)\n"
        "if r == 0: print(json.dumps({'metric': 'm', 'n_gpus': w, 'argv': sys.argv[1:]}))\n")
    drv = tmp_path / "drv.py"
    drv.write_text(
        "import sys, importlib.util\n"
        f"spec = importlib.util.spec_from_file_location('bench_mod', {os.path.join(ROOT, 'bench.py')!r})\n"
        "b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
        f"b.__file__ = {str(fake)!r}\n"
        "b.self_launch(2)\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    ok = subprocess.run([sys.executable, str(drv), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, env=env, timeout=300)
    assert ok.returncode == 0, ok.stderr[-2000:]
    line = [l for l in ok.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1
    rec = json.loads(line[0])
    assert rec["n_gpus"] == 2 and rec["argv"] == ["--gpus", "2", "--steps", "1"]
    bad = subprocess.run([sys.executable, str(drv), "--gpus", "2", "--fail"], capture_output=True, text=True, env=env, timeout=300)
    assert bad.returncode != 0 and "rank failed" in bad.stderr


def test_library_has_no_packed_fp32_valu_instructions():
    """Build hygiene that a correctness finding hangs on: with another process sharing the GPU, compiler-formed
    packed fp32 VALU ops (v_pk_add_f32 with an op_sel broadcast, in layernorm_fwd_kernel) dropped their subtraction in lanes 48-63
    of one register once per ~50-100 launches.  The library is built with -fno-slp-vectorize -fno-vectorize; this test disassembles
    the gfx950 code objects embedded in the .so and checks that none of those instructions (and no LDS-crossbar shuffle) is left."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_scan
    if not os.path.exists(isa_scan.OBJDUMP):
        pytest.skip("llvm-objdump not found")
    res, lines = isa_scan.count([r"v_pk_[a-z0-9]+_f32", r"ds_bpermute_b32", r"ds_swizzle_b32", r"v_mfma_\w+"])
    assert lines > 100000, "disassembly looks empty"
    assert res[r"v_mfma_\w+"] > 1000                      # the scan does see the kernels
    assert res[r"v_pk_[a-z0-9]+_f32"] == 0, res
    assert res[r"ds_bpermute_b32"] == 0 and res[r"ds_swizzle_b32"] == 0, res


def test_bench_default_protocol_meets_survey_8d(monkeypatch):
    """>= 20 timed steps behind >= 5 warm-up steps; bench.py's defaults (plus the settle phase that keeps the clock ramp of a cold
    start out of the timed region) and the N = 1 default the driver relies on."""
    import sys
    sys.path.insert(0, ROOT) if ROOT not in sys.path else None
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert a.gpus == 1 and a.steps >= 20 and a.warmup >= 5 and 0 < a.settle_seconds <= 5
    assert a.batch == 4 and a.frames == 16 and a.size == 256 and a.dtype == "bf16" and a.workload == "vae"      # config C3
