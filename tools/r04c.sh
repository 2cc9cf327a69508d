cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04c && timeout -k 10 300 python tools/pp_clock.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04c/pp_clock.txt
