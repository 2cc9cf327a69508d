#!/usr/bin/env python3
"""Which TunableOp-selected library GEMM produced the non-finite loss recorded in round 1 (gpurun_out/tun.err)?

    PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=0 PYTORCH_TUNABLEOP_FILENAME=tools/tunableop_r01_results.csv \
        python tools/tunable_probe.py

Replays every Linear product of the production step (forward addmm with bias, input gradient mm against W^T) with the solutions
round 1's tuning run selected (tools/tunableop_r01_results.csv) and compares each with an fp32 product of the same bf16 operands.
"""
import json
import torch

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
M = 16384
rows = []
for k, n in [(768, 768), (768, 1536), (512, 768), (1536, 768), (768, 96), (96, 768), (768, 3072), (96, 1), (256, 1)]:
    m = M if k != 256 else 64
    x = (torch.randn((m, k), generator=g)).to(dev, torch.bfloat16)
    w = (torch.randn((k, n), generator=g) * k ** -0.5).to(dev, torch.bfloat16)
    b = torch.randn((n,), generator=g).to(dev, torch.bfloat16)
    dy = torch.randn((m, n), generator=g).to(dev, torch.bfloat16)
    ref = (x.float() @ w.float() + b.float())
    got = torch.addmm(b, x, w).float()
    rows.append({"op": "addmm", "m": m, "k": k, "n": n, "finite": bool(torch.isfinite(got).all()),
                 "max_err": float((got - ref).abs().max()), "ref_max": float(ref.abs().max())})
    ref = dy.float() @ w.float().t()
    got = torch.mm(dy, w.t()).float()
    rows.append({"op": "mm(dy, W^T)", "m": m, "k": n, "n": k, "finite": bool(torch.isfinite(got).all()),
                 "max_err": float((got - ref).abs().max()), "ref_max": float(ref.abs().max())})
torch.cuda.synchronize()
for r in rows:
    r["bad"] = (not r["finite"]) or r["max_err"] > 0.05 * max(r["ref_max"], 1.0)
    print(json.dumps(r))
