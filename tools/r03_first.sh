#!/bin/bash
# Round-3 first evidence run: parity test of the benchmarked model, the two reduction probes, a baseline line, the MFMA-busy pass.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03a
mkdir -p $O
cd $R
echo "== parity r3"; timeout -k 10 900 python -m pytest tests/test_gpu_parity_r3.py -x -q -s -m gpu > $O/parity_r3.log 2>&1; echo "rc $?"; tail -5 $O/parity_r3.log
echo "== memset node probe"; timeout -k 10 300 python tools/memset_node_probe.py 60 > $O/memset_probe.log 2>&1; echo "rc $?"; cat $O/memset_probe.log | tail -8
echo "== reduce history probe"; timeout -k 10 400 python tools/reduce_history_probe.py 3 $O/step_graph.dot > $O/reduce_probe.log 2>&1; echo "rc $?"; tail -40 $O/reduce_probe.log
gzip -f $O/step_graph.dot 2>/dev/null
echo "== bench"; timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_line.json 2> $O/bench.err; echo "rc $?"; cut -c1-600 $O/bench_line.json
cd /tmp && export TMPDIR=/tmp
echo "== PMC MFMA busy"; timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- python3 $R/bench.py --no-graph --steps 2 --warmup 1 --settle-seconds 0 --no-cpu-baseline --no-kernel-timing > $O/pmc_mfma.log 2>&1; echo "rc $?"
cd $R
python tools/pmc_mfma_sum.py "$O/pmc_mfma/*/*counter_collection.csv" $O/pmc_mfma_busy.txt $O/pmc_mfma_families.json > /dev/null; head -30 $O/pmc_mfma_busy.txt
python - <<PY
import csv, glob, collections
csv.field_size_limit(1 << 30)
seen = collections.Counter()
for f in glob.glob("$O/pmc_mfma/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and ("reduce_kernel" in r["Kernel_Name"] or "fillBuffer" in r["Kernel_Name"]):
            seen[(r["Kernel_Name"][:90], r["Grid_Size"], r["Workgroup_Size"])] += 1
for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(v, k)
PY
rm -rf $O/pmc_mfma
