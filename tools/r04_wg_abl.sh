#!/bin/bash
# timing-only variants of conv3d_wgrad_bf16_kernel (conv3d_bf16.hip -DWG_ABL=bits), built HERE; then: gpurun -- 'bash tools/r04_wg_abl.sh run'
cd $(dirname $0)/../video_vae_amd/csrc
if [ "$1" != "run" ]; then
  for b in 0 1 2 4 3 5 6 7; do
    hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-slp-vectorize -fno-vectorize -DWG_ABL=$b -c conv3d_bf16.hip -o build/wg_abl.o || exit 1
    hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v "conv3d_bf16.o") -L/opt/rocm/lib -lhipblaslt -Wl,-rpath,/opt/rocm/lib -o build/libvvae_hip_wgabl$b.so || exit 1
    rm build/wg_abl.o
  done
  exit 0
fi
cd ../..
mkdir -p gpurun_out/r04d
for b in 0 1 2 4 3 5 6 7 0; do
  echo "== WG_ABL=$b (1 no MFMA, 2 no fragment reads, 4 no staging behind the first planes)" | tee -a gpurun_out/r04d/wg_abl.txt
  VVAE_AB_LIB=video_vae_amd/csrc/build/libvvae_hip_wgabl$b.so timeout -k 10 200 python tools/wgrad_bench.py 0 0 2>&1 | grep -v amdgpu.ids | sed 's/ e[0-9a-z.+-]*$//' | tee -a gpurun_out/r04d/wg_abl.txt
done
