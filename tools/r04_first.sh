#!/bin/bash
# Round 4, first GPU call: parity suite with the tightened bars, the selection_layer1.bias attribution, GEMM census of the replayed step.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04a
mkdir -p $O
cd $R
stop_if_killed() { if [ $1 -ge 124 ]; then echo "step killed ($1): stopping"; exit $1; fi; }
echo "== pp bench"; timeout -k 10 240 python tools/pp_bench.py > $O/pp_bench.txt 2>&1; rc=$?; cat $O/pp_bench.txt | grep -v amdgpu.ids | cut -c1-400; stop_if_killed $rc
echo "== gpu tests"; timeout -k 10 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_parity_r2.py::test_production_step_survives_unsynchronised_bursts > $O/tests.log 2>&1; rc=$?; grep -E 'passed|failed|FAILED|ERROR' $O/tests.log | tail -15; stop_if_killed $rc
echo "== attribution"; timeout -k 10 300 python tools/r04_sel1_attribution.py > $O/sel1_attribution.txt 2> $O/sel1.err; rc=$?; cat $O/sel1_attribution.txt; stop_if_killed $rc
echo "== nt bench"; timeout -k 10 200 python tools/nt_bench.py > $O/nt_bench.txt 2>&1; rc=$?; tail -14 $O/nt_bench.txt; stop_if_killed $rc
cd /tmp && export TMPDIR=/tmp
echo "== kernel trace"; timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timing --no-also --steps 10 --warmup 2 --settle-seconds 0.5 > $O/trace.log 2>&1; rc=$?; stop_if_killed $rc
cd $R
python tools/r04_gemm_census.py $O/trace > $O/gemm_census.txt 2>&1; cat $O/gemm_census.txt | cut -c1-200
python tools/r04_gemm_census.py $O/trace layernorm sattn tattn adam conv3d gn_silu > $O/other_census.txt 2>&1
rm -rf $O/trace
