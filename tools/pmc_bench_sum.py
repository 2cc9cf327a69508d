import csv, glob, collections, sys
pat = sys.argv[1]
out = sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(pat):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, cs in agg.items():
    rows.append((sum(cs.get("FETCH_SIZE", [0])) * 2 + sum(cs.get("WRITE_SIZE", [0])), k, cs))
rows.sort(reverse=True)
def avg(cs, n):
    return sum(cs[n]) / len(cs[n]) if n in cs else float("nan")
with open(out, "w") as o:
    o.write("# rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE TCC_HIT_sum TCC_MISS_sum) over `python bench.py --no-graph --steps 2 --warmup 1`\n")
    o.write("# per-launch averages; FETCH_SIZE/WRITE_SIZE in KB as reported.  gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide\n")
    o.write("# coalesced read -> hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md, HBM section)\n")
    o.write("kernel | launches | FETCH_SIZE_KB | WRITE_SIZE_KB | hbm_MB_corrected | TCC_HIT | TCC_MISS\n")
    for _, k, cs in rows[:45]:
        f, w = avg(cs, "FETCH_SIZE"), avg(cs, "WRITE_SIZE")
        o.write(f"{k} | {len(cs.get('FETCH_SIZE', []))} | {f:.1f} | {w:.1f} | {(2 * f + w) * 1024 / 1e6:.2f} | {avg(cs, 'TCC_HIT_sum'):.0f} | {avg(cs, 'TCC_MISS_sum'):.0f}\n")
print(open(out).read()[:4500])
