#!/bin/bash
# full GPU suite + default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r03g}
mkdir -p $O
cd $R
echo "== full GPU suite"; timeout -k 10 1000 python -m pytest tests -q -m gpu -x > $O/gputests.log 2>&1; echo "rc $?"; tail -6 $O/gputests.log | cut -c1-300
echo "== bench"; timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_line.json 2> $O/bench.err; echo "rc $?"; python - <<PY
import json
d=json.loads(open("$O/bench_line.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_ms"], d["conv_stack"])
for k in d["kernels"]: print(k)
PY
