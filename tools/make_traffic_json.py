#!/usr/bin/env python3
"""profiles/rNN_traffic.json from the per-family table of tools/pmc_bench_sum.py (the FETCH_SIZE / WRITE_SIZE passes over bench.py).

    python tools/make_traffic_json.py gpurun_out/pmc_families.json profiles/r03_traffic.json [gpurun_out/pmc_mfma_families.json]

With the third argument (tools/pmc_mfma_sum.py) every entry also carries ``mfma_busy``, the matrix pipes' busy fraction of the launch.

Each entry carries the sha256 of the .hip files its kernel is built from; bench.py refuses the figure when those sources change."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# bench.py kernel tag -> (kernel families of the PMC table, source files)
KERNELS = {
    "conv3d_bf16_roll_kernel|conv3d_bf16_deep_kernel": (["conv3d_bf16_roll_kernel", "conv3d_bf16_deep_kernel", "conv3d_bf16_kernel"], ["conv3d_bf16.hip", "common.hpp"]),
    "conv3d_wgrad": (["conv3d_wgrad_bf16_kernel", "wgrad_reduce_kernel"], ["conv3d_bf16.hip", "common.hpp"]),
    "gemm_tn256_grouped_kernel": (["gemm_tn256_grouped_kernel"], ["gemm_tn256.hip", "common.hpp"]),
    "tattn_fwd_fast": (["tattn16_fwd_mfma"], ["attn_temporal_mfma.hip", "common.hpp"]),
    "tattn_bwd_fast": (["tattn16_bwd_mfma"], ["attn_temporal_mfma.hip", "common.hpp"]),
    "sattn_fwd_kernel": (["sattn_fwd_kernel"], ["attn_spatial.hip", "common.hpp"]),
    "sattn_bwd_kernel": (["sattn_bwd_kernel"], ["attn_spatial.hip", "common.hpp"]),
    "layernorm_fwd_kernel": (["layernorm_fwd_kernel"], ["layernorm.hip", "common.hpp"]),
    "layernorm_bwd_kernel": (["layernorm_bwd_kernel"], ["layernorm.hip", "common.hpp"]),
    "gemm_nt_kernel": (["gemm_nt_kernel"], ["gemm_nt.hip", "common.hpp"]),
    "gemm_pp_kernel": (["gemm_pp_kernel"], ["gemm_pp.hip", "common.hpp"]),
}


def sha(files):
    h = hashlib.sha256()
    for name in files:
        with open(os.path.join(ROOT, "video_vae_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


fam = json.load(open(sys.argv[1]))
mf = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else {}
out = {}
for tag, (names, files) in KERNELS.items():
    tot = sum(fam[n]["hbm_bytes_total"] for n in names if n in fam)
    n = sum(fam[x]["launches"] for x in names if x in fam)
    if n:
        out[tag] = {"hbm_bytes_per_launch": tot / n, "launches": n, "families": names, "source_files": files, "source_sha256": sha(files),
                    "how": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE passes over `bench.py --no-graph --steps 2 --warmup 1`; "
                           "(2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch (gfx950: FETCH_SIZE counts 64 B per 128-B request)"}
    if tag in out:
        busy, act = sum(mf[x]["busy_cycles"] for x in names if x in mf), sum(mf[x]["gui_active"] for x in names if x in mf)
        if act:
            out[tag]["mfma_busy"] = busy / (act / 8.0 * 1024.0)
            out[tag]["mfma_busy_how"] = ("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass over the same command; busy cycles / "
                                         "(GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs), all launches of the family together")
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 2) for k, v in out.items()}, indent=1))
