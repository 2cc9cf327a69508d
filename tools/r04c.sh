cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04c && PYTHONPATH=tools timeout -k 10 300 python tools/pp_ablation.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04c/pp_ablation.txt
