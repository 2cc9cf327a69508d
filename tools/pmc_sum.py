import csv, glob, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if not any(x in k for x in ("gemm", "Cijk", "conv3d", "layernorm", "tattn", "sattn")):
        continue
    print(k)
    for c, v in sorted(cs.items()):
        v2 = v[len(v) // 2:]          # skip warm-up half
        print(f"   {c:28s} n={len(v):3d} avg {sum(v2)/len(v2):16.1f}")
