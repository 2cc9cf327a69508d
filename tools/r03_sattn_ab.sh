#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
echo "== attention tests"; timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "spatial" 2>&1 | tail -2
for i in 1 2 3; do
echo "NEW  $(timeout -k 10 100 python tools/sattn_bench.py 2>&1 | grep sattn)"
echo "BASE $(VVAE_AB_LIB=$R/video_vae_amd/csrc/build/libvvae_hip_base.so timeout -k 10 100 python tools/sattn_bench.py 2>&1 | grep sattn)"
done
