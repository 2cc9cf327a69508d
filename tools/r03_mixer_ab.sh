#!/bin/bash
# A/B of the patch-mixer rolling configuration: new build vs video_vae_amd/csrc/build/libvvae_hip_base.so
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r03r}
mkdir -p $O
cd $R
echo "== conv tests"; timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "conv or unet or mixer or roll" > $O/conv_tests.log 2>&1; echo "rc $?"; tail -3 $O/conv_tests.log
for i in 1 2; do
echo "== NEW"; timeout -k 10 200 python tools/mixer_bench.py 2>&1 | grep mixer
echo "== BASE"; VVAE_AB_LIB=$R/video_vae_amd/csrc/build/libvvae_hip_base.so timeout -k 10 200 python tools/mixer_bench.py 2>&1 | grep mixer
done
