"""Per-shape census of the GEMM launches of the replayed train step from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --no-cpu-baseline --no-kernel-timing --no-also ...
    python tools/r04_gemm_census.py DIR [pattern ...]
Groups dispatches by (kernel name, grid, workgroup) -- one group per product shape of a kernel -- and prints launches per step, mean and
median duration, total ms per step.  Steps are counted by adam_clip_kernel launches.  Default patterns: the GEMM families."""
import csv
import glob
import statistics
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
d = sys.argv[1]
pats = sys.argv[2:] or ["Cijk_", "gemm_nt", "gemm_tn", "gemm_pp"]
files = glob.glob(d + "/*/*kernel_trace.csv") + glob.glob(d + "/*kernel_trace.csv")
assert files, "no kernel_trace.csv under " + d
groups, steps = defaultdict(list), 0
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "adam_clip_kernel" in name:
            steps += 1
        if any(p in name for p in pats):
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            wg = int(r["Workgroup_Size_X"]) if "Workgroup_Size_X" in r else int(r.get("Workgroup_Size", 0))
            grid = int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))
            short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            short = short if len(short) <= 100 else short[:60] + ".." + short[-38:]
            groups[(short, grid // max(wg, 1), wg)].append(dur)
steps = max(steps, 1)
tot = 0.0
print(f"# steps {steps}; per group: launches/step | mean us | median us | ms/step | workgroups x threads | kernel")
for k, v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
    ms = sum(v) / steps / 1e3
    tot += ms
    print(f"{len(v) / steps:7.1f} | {statistics.mean(v):8.1f} | {statistics.median(v):8.1f} | {ms:7.3f} | {k[1]:5d} x {k[2]:4d} | {k[0]}")
print(f"# total {tot:.3f} ms/step over {sum(len(v) for v in groups.values()) / steps:.1f} launches/step")
