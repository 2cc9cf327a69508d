#!/bin/bash
# LDS conflict / VALU / MFMA counters of the spatial attention kernels:  bash tools/r03_sattn_pmc.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  d=$O/$(echo $set | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python3 $R/tools/sattn_bench.py > $d.log 2>&1 || echo "FAILED $set"
done
cd $R
python3 - <<PY
import csv, glob, collections
csv.field_size_limit(1 << 30)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$O/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "sattn" not in k: continue
        name = "fwd" if "fwd" in k else "bwd"
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] in ("SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES"): n[(name, r["Counter_Name"])] += 1
for name, c in acc.items():
    print(name, {k: round(v / max(1, n[(name, "SQ_LDS_BANK_CONFLICT")] or n[(name, "SQ_INSTS_VALU")] or n[(name, "SQ_WAVE_CYCLES")]), 1) for k, v in sorted(c.items())})
PY
