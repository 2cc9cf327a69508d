#!/bin/bash
# a batch of whole-step A/Bs (tools/ab_hook.py), one process each:  bash tools/r03_ab_batch.sh <outdir>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r03p}
mkdir -p $O
cd $R
run() { echo "== $*"; timeout -k 10 300 python tools/ab_hook.py "$@" r=4 2>&1 | grep "ms/step" | cut -c1-200; }
run py:video_vae_amd.ops.GROUP_TILES 512 448 640 768 > $O/ab.log 2>&1
run vvae_gemm_nt_prefetch 3 2 4 5 >> $O/ab.log 2>&1
run vvae_layernorm_config 384 320 448 >> $O/ab.log 2>&1
run vvae_conv3d_deep_config 1 0 >> $O/ab.log 2>&1
cat $O/ab.log
