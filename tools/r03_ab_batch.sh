#!/bin/bash
# a batch of whole-step A/Bs (tools/ab_hook.py), one process each:  bash tools/r03_ab_batch.sh <outdir>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r03p}
mkdir -p $O
cd $R
run() { echo "== $*"; timeout -k 10 300 python tools/ab_hook.py "$@" r=4 2>&1 | grep "ms/step" | cut -c1-200; }
run py:video_vae_amd.ops.GROUP_MIN_TILES 128 64 32 256 > $O/ab.log 2>&1
run py:video_vae_amd.ops.FOLD_MAX 64 32 16 >> $O/ab.log 2>&1
cat $O/ab.log
