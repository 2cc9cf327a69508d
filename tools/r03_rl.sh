#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03rl
mkdir -p $O
cd $R
echo "== tests"; timeout -k 10 700 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity_r3.py tests/test_gpu_train.py tests/test_gpu_model.py tests/test_perceptual.py -q -m gpu -x -k "rl or loss or flavour or train or perceptual" > $O/tests.log 2>&1; echo "rc $?"; tail -3 $O/tests.log | cut -c1-200
echo "== bench rl"; timeout -k 10 300 python bench.py --no-cpu-baseline --flavour rl > $O/bench.json 2> $O/bench.err; echo "rc $?"
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"].get("graph_nodes"), d["config"].get("flavour"))
PY
