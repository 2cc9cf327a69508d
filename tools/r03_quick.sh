#!/bin/bash
# a pytest selection + the default bench line:  bash tools/r03_quick.sh <outdir> "<pytest -k expression>" [test files...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r03q}
K=${2:-conv}
shift 2
F=${@:-tests}
mkdir -p $O
cd $R
echo "== tests -k '$K'"; timeout -k 10 700 python -m pytest $F -q -m gpu -x -k "$K" > $O/tests.log 2>&1; echo "rc $?"; tail -4 $O/tests.log | cut -c1-300
echo "== bench"; timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_line.json 2> $O/bench.err; echo "rc $?"; python - <<PY
import json
d=json.loads(open("$O/bench_line.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"].get("graph_nodes"), d["roofline"]["frac"], d["roofline"]["avg_ms"], d["conv_stack"]["ms_per_step"], d["timing"]["cold"]["ms_per_step_median"])
PY
