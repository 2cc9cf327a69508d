#!/bin/bash
# kernel-only times of ops.encoder_head under rocprofv3:  bash tools/r03_head_prof.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03k
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/tools/head_bench.py 50 > $O/prof.log 2>&1 || echo FAILED
cd $R
python3 - <<PY
import csv, glob
for f in glob.glob("$O/prof/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("encoder_head", "sum_rows", "fold_rows")):
            print(r["Name"][:60], r["Calls"], "avg us", float(r["AverageNs"]) / 1e3, "min", float(r["MinNs"]) / 1e3, "max", float(r["MaxNs"]) / 1e3)
PY
rm -rf $O/prof
