#!/bin/bash
# builds timing variants of gemm_tn256.hip HERE (CPU container), then: gpurun -- 'bash tools/r04_tn_abl.sh run'
cd $(dirname $0)/../video_vae_amd/csrc
if [ "$1" != "run" ]; then
  for a in 2 3 4 5; do
    hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-slp-vectorize -fno-vectorize -DTN_ALATE=$a -DTN_STAMPS -c gemm_tn256.hip -o build/tn_abl.o || exit 1
    hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v "gemm_tn256.o") -L/opt/rocm/lib -lhipblaslt -Wl,-rpath,/opt/rocm/lib -o build/libvvae_hip_tnal$a.so || exit 1
    rm build/tn_abl.o
  done
  exit 0
fi
cd ../..
mkdir -p gpurun_out/r04c
timeout -k 10 200 python tools/tn_pp_check.py 2>&1 | grep -v amdgpu.ids | head -4
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -k "gemm_tn" 2>&1 | tail -2
for a in 2 3 4 5 3; do
  VVAE_AB_LIB=video_vae_amd/csrc/build/libvvae_hip_tnal$a.so timeout -k 10 120 python tools/tn_ablation.py "A-early-$a" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04c/tn_alate.txt
done
