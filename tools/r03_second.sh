#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03b
mkdir -p $O
cd $R
echo "== parity r3 + train driver tests"; timeout -k 10 900 python -m pytest tests/test_gpu_parity_r3.py tests/test_gpu_train.py "tests/test_gpu_model.py::test_sigterm_checkpoints_and_exits" -q -s -m gpu > $O/tests.log 2>&1; echo "rc $?"; grep -n "passed\|failed\|worst\|needed more\|Error" $O/tests.log | cut -c1-1500 | tail -30
echo "== memset node probe"; timeout -k 10 300 python tools/memset_node_probe.py 40 > $O/memset_probe.log 2>&1; echo "rc $?"; grep "part" $O/memset_probe.log | cut -c1-400
echo "== train driver on the production shape"; timeout -k 10 400 python -m video_vae_amd.train --per_device_batch_size 4 --max_frames 16 --flavour model --steps 90 --log_every 30 > $O/train_prod.log 2>&1; echo "rc $?"; grep "captured\|summary\|Step 60" $O/train_prod.log | cut -c1-300
echo "== bench"; timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_line.json 2> $O/bench.err; echo "rc $?"; cut -c1-400 $O/bench_line.json
echo "== bench eager"; timeout -k 10 300 python bench.py --no-cpu-baseline --eager --steps 10 > $O/bench_eager_line.json 2> $O/bench_eager.err; echo "rc $?"; cut -c1-300 $O/bench_eager_line.json
