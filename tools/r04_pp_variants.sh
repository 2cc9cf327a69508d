#!/bin/bash
# builds variants of gemm_pp.hip HERE (CPU container), then: gpurun -- 'bash tools/r04_pp_variants.sh run'
cd $(dirname $0)/../video_vae_amd/csrc
VARS="0:0 1:0 1:1 1:2 0:2 1:3"
if [ "$1" != "run" ]; then
  for v in $VARS; do
    l=${v%%:*}; c=${v##*:}
    hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-slp-vectorize -fno-vectorize -DPP_LATE=$l -DPP_CDMA=$c -c gemm_pp.hip -o build/pp_var.o || exit 1
    hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v "gemm_pp.o") -L/opt/rocm/lib -lhipblaslt -Wl,-rpath,/opt/rocm/lib -o build/libvvae_hip_ppv_${l}_${c}.so || exit 1
    rm build/pp_var.o
  done
  exit 0
fi
cd ../..
mkdir -p gpurun_out/r04d
timeout -k 10 300 python tools/pp_bench.py 2>&1 | grep -v amdgpu.ids | grep "mismatches" | head -8
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -k "gemm_pp or gemm_nt" 2>&1 | tail -2
for v in $VARS 0:0 1:2; do
  l=${v%%:*}; c=${v##*:}
  VVAE_AB_LIB=video_vae_amd/csrc/build/libvvae_hip_ppv_${l}_${c}.so timeout -k 10 200 python tools/pp_variants.py "late$l-cd$c" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04d/pp_variants.txt
done
