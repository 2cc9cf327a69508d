#!/bin/bash
# Round-2 evidence run (one 1-GPU box):  bash tools/r02_profiles.sh   -> files under gpurun_out/r02/
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# the PMC passes first: bench.py quotes `traffic` from profiles/r02_traffic.json only while the kernel sources match its hashes
echo "== PMC FETCH_SIZE"; timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --no-graph --steps 2 --warmup 1 --settle-seconds 0 --no-cpu-baseline > $O/pmc_fetch.log 2>&1 || echo FAILED
echo "== PMC WRITE_SIZE"; timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --no-graph --steps 2 --warmup 1 --settle-seconds 0 --no-cpu-baseline > $O/pmc_write.log 2>&1 || echo FAILED
cd $R
python tools/pmc_bench_sum.py "$O/pmc_*/*/*counter_collection.csv" $O/pmc_hbm_traffic.txt $O/pmc_families.json > /dev/null
python tools/make_traffic_json.py $O/pmc_families.json $O/traffic.json
cp $O/traffic.json $R/profiles/r02_traffic.json
echo "== default bench line (with cpu_baseline)"; timeout -k 10 400 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err || echo FAILED
echo "== input pipeline"; timeout -k 10 300 python bench.py --no-cpu-baseline --with-input-pipeline > $O/bench_input_pipeline_line.json 2> $O/bench_input_pipeline.err || echo FAILED
echo "== C5 shape"; timeout -k 10 300 python bench.py --no-cpu-baseline --batch 2 --frames 32 > $O/bench_c5_b2_t32_line.json 2> $O/bench_c5.err || echo FAILED
echo "== single-rank RCCL rehearsal"; timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-ddp --no-cpu-baseline > $O/bench_force_ddp_line.json 2> $O/bench_force_ddp.err || echo FAILED
cd /tmp && export TMPDIR=/tmp
echo "== kernel trace"; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 $R/bench.py --no-cpu-baseline > $O/prof_default.log 2>&1 || echo FAILED
cd $R
python tools/prof_summary.py $O/prof_default 60 > $O/bench_default_per_step_summary.txt
cp $(ls $O/prof_default/*/*kernel_stats.csv | head -1) $O/bench_default_kernel_stats.csv
rm -rf $O/prof_default $O/pmc_fetch $O/pmc_write
for f in bench_default_line bench_input_pipeline_line bench_c5_b2_t32_line bench_force_ddp_line; do python - <<PY
import json
try:
    d = json.loads(open("$O/$f.json").read().strip().splitlines()[-1])          # RCCL prints its version banner on stdout first
    print("$f", round(d["value"], 1), "frames/s", round(d["ms_per_step"], 2), "ms/step", d.get("rccl_ranks"), (d.get("roofline") or {}).get("frac"))
except Exception as e:
    print("$f", "unreadable:", e)
PY
done
head -25 $O/bench_default_per_step_summary.txt | cut -c1-140
