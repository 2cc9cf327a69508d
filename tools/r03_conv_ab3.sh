#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r03p}
mkdir -p $O
cd $R
echo "== conv tests"; timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "conv or unet or wgrad or roll or mixer" > $O/conv_tests.log 2>&1; echo "rc $?"; tail -3 $O/conv_tests.log
echo "== fwd/dgrad NEW"; timeout -k 10 300 python tools/conv_bench.py 1 0 > $O/conv_new.log 2>&1; grep -v amdgpu $O/conv_new.log | cut -c1-120 | tail -12
echo "== fwd/dgrad BASE"; VVAE_AB_LIB=$R/video_vae_amd/csrc/build/libvvae_hip_base.so timeout -k 10 300 python tools/conv_bench.py 1 0 > $O/conv_base.log 2>&1; grep -v amdgpu $O/conv_base.log | cut -c1-120 | tail -12
echo "== wgrad NEW"; timeout -k 10 300 python tools/wgrad_bench.py 0 0 > $O/wgrad_new.log 2>&1; grep -v amdgpu $O/wgrad_new.log | cut -c1-120
echo "== wgrad BASE"; VVAE_AB_LIB=$R/video_vae_amd/csrc/build/libvvae_hip_base.so timeout -k 10 300 python tools/wgrad_bench.py 0 0 > $O/wgrad_base.log 2>&1; grep -v amdgpu $O/wgrad_base.log | cut -c1-120
