#!/bin/bash
# Round-4 evidence run (one 1-GPU box):  bash tools/r04_profiles.sh   -> files under gpurun_out/r04/ (copy what is to be judged to profiles/)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-graph --steps 2 --warmup 1 --settle-seconds 0 --no-cpu-baseline --no-kernel-timing --no-also"
# the PMC passes first: bench.py quotes `traffic` / `mfma_busy` from profiles/r04_traffic.json only while the kernel sources match its hashes
echo "== PMC FETCH_SIZE"; timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1 || echo FAILED
echo "== PMC WRITE_SIZE"; timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1 || echo FAILED
echo "== PMC MFMA busy"; timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- $B > $O/pmc_mfma.log 2>&1 || echo FAILED
cd $R
python tools/pmc_bench_sum.py "$O/pmc_[fw]*/*/*counter_collection.csv" $O/pmc_hbm_traffic.txt $O/pmc_families.json > /dev/null
python tools/pmc_mfma_sum.py "$O/pmc_mfma/*/*counter_collection.csv" $O/pmc_mfma_busy.txt $O/pmc_mfma_families.json > /dev/null
python tools/make_traffic_json.py $O/pmc_families.json $O/traffic.json $O/pmc_mfma_families.json
cp $O/traffic.json $R/profiles/r04_traffic.json
# which framework reductions / memset kernels one eager step launches (grid, workgroup): evidence for DESIGN section 3
python - > $O/reduce_and_memset_launches.txt <<PY
import csv, glob, collections
csv.field_size_limit(1 << 30)
seen = collections.Counter()
for f in glob.glob("$O/pmc_mfma/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and ("at::native::reduce_kernel" in r["Kernel_Name"] or "fillBuffer" in r["Kernel_Name"]):
            seen[(r["Kernel_Name"][:110], int(r["Grid_Size"]) // int(r["Workgroup_Size"]), r["Workgroup_Size"])] += 1
print("# launches over 3 eager steps | kernel | workgroups | workgroup size")
for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(v, "|", k[0], "|", k[1], "|", k[2])
PY
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_mfma
echo "== default bench line (with cpu_baseline)"; timeout -k 10 600 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err || echo FAILED
echo "== eager line"; timeout -k 10 300 python bench.py --no-cpu-baseline --no-also --eager --steps 10 > $O/bench_eager_line.json 2> $O/bench_eager.err || echo FAILED
echo "== train.py on the same shape"; timeout -k 10 400 python -m video_vae_amd.train --per_device_batch_size 4 --max_frames 16 --flavour model --steps 90 --log_every 30 > $O/train_prod.log 2>&1 || echo FAILED
echo "== input pipeline"; timeout -k 10 300 python bench.py --no-cpu-baseline --no-also --with-input-pipeline > $O/bench_input_pipeline_line.json 2> $O/bench_input_pipeline.err || echo FAILED
echo "== C5 shape"; timeout -k 10 300 python bench.py --no-cpu-baseline --no-also --batch 2 --frames 32 > $O/bench_c5_b2_t32_line.json 2> $O/bench_c5.err || echo FAILED
echo "== single-rank RCCL rehearsal, 1 + 9 graphs"; timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-ddp --enc-segments 9 --no-cpu-baseline > $O/bench_force_ddp_line.json 2> $O/bench_force_ddp.err || echo FAILED
echo "== single-rank RCCL rehearsal, 1 + 3 graphs"; timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --force-ddp --enc-segments 3 --no-cpu-baseline > $O/bench_force_ddp_seg3_line.json 2> $O/bench_force_ddp_seg3.err || echo FAILED
echo "== single-rank RCCL rehearsal, bf16 gradient all-reduce"; timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29535 bench.py --gpus 1 --force-ddp --grad-dtype bf16 --no-cpu-baseline > $O/bench_force_ddp_bf16_line.json 2> $O/bench_force_ddp_bf16.err || echo FAILED
echo "== rl flavour line"; timeout -k 10 300 python bench.py --no-cpu-baseline --no-also --flavour rl --steps 20 > $O/bench_rl_flavour_line.json 2> $O/bench_rl.err || echo FAILED
cd /tmp && export TMPDIR=/tmp
echo "== kernel trace"; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 $R/bench.py --no-cpu-baseline --no-also > $O/prof_default.log 2>&1 || echo FAILED
cd $R
python tools/prof_summary.py $O/prof_default 70 > $O/bench_default_per_step_summary.txt
cp $(ls $O/prof_default/*/*kernel_stats.csv | head -1) $O/bench_default_kernel_stats.csv
rm -rf $O/prof_default
for f in bench_default_line bench_eager_line bench_input_pipeline_line bench_c5_b2_t32_line bench_force_ddp_line bench_force_ddp_seg3_line bench_force_ddp_bf16_line bench_rl_flavour_line; do python - <<PY
import json
try:
    d = json.loads(open("$O/$f.json").read().strip().splitlines()[-1])          # RCCL prints its version banner on stdout first
    r = d.get("roofline") or {}
    print("$f", round(d["value"], 1), "frames/s", round(d["ms_per_step"], 2), "ms/step", d.get("rccl_ranks"), r.get("frac"), r.get("mfma_busy"), (d.get("conv_stack") or {}).get("ms_per_step"))
except Exception as e:
    print("$f", "unreadable:", e)
PY
done
grep "captured\|summary" $O/train_prod.log | cut -c1-200
head -30 $O/bench_default_per_step_summary.txt | cut -c1-140
