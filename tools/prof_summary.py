import csv, sys, glob
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
steps = [int(r['Calls']) for r in rows if 'adam_clip' in r['Name']][0]
tot = sum(float(r['TotalDurationNs']) for r in rows) / steps / 1e6
print("steps", steps, "total ms/step %.2f" % tot)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for r in rows[:n]:
    print(f"{float(r['TotalDurationNs'])/steps/1e6:7.3f} ms  {int(r['Calls'])/steps:6.1f} x {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:100]}")
