#!/bin/bash
# kernel census of the rl flavour (the reference driver's own model): gpurun -- 'bash tools/r04_rl_census.sh'
O=gpurun_out/r04e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rl -- python3 bench.py --no-cpu-baseline --no-also --no-kernel-timing --flavour rl --steps 20 --warmup 3 --settle-seconds 0 > $O/rl_prof.log 2>&1 || { echo FAILED; tail -5 $O/rl_prof.log; exit 1; }
f=$(find $O/prof_rl -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' > $O/rl_kernel_census.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --flavour rl --steps 20 --warmup 3 (all launches of the process: warm-up, capture, cold + settled legs)")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:70]:
    print(f"{float(r['TotalDurationNs']) / tot * 100:5.1f} %  {int(r['Calls']):7d} x {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:140]}")
PY
find $O/prof_rl -name "*.csv" -size +3M -delete
grep '"metric"' $O/rl_prof.log | cut -c1-300
