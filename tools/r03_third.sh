#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r03i}
mkdir -p $O
cd $R
echo "== train driver tests"; timeout -k 10 600 python -m pytest tests/test_gpu_train.py "tests/test_gpu_model.py::test_sigterm_checkpoints_and_exits" -q -m gpu -x > $O/tests.log 2>&1; echo "rc $?"; tail -4 $O/tests.log | cut -c1-300; grep -n "Error" -B2 -A8 $O/tests.log | head -60 | cut -c1-300
echo "== train driver on the production shape"; timeout -k 10 400 python -m video_vae_amd.train --per_device_batch_size 4 --max_frames 16 --flavour model --steps 90 --log_every 30 > $O/train_prod.log 2>&1; echo "rc $?"; grep "captured\|summary\|Step 60" $O/train_prod.log | cut -c1-300
echo "== bench"; timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_line.json 2> $O/bench.err; echo "rc $?"; python - <<PY
import json
d=json.loads(open("$O/bench_line.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_ms"], d["conv_stack"])
PY
