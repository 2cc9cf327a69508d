"""Per-kernel HBM traffic table from the FETCH_SIZE and WRITE_SIZE passes of `rocprofv3 --pmc` over bench.py.
usage: python tools/pmc_bench_sum.py '<glob of *counter_collection.csv>' <out.txt> [<families.json>]"""
import csv, glob, collections, json, re, sys
pat, out = sys.argv[1], sys.argv[2]
fam_out = sys.argv[3] if len(sys.argv) > 3 else None
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(pat):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:150]][r["Counter_Name"]].append(float(r["Counter_Value"]))


def base(name):
    m = re.search(r"([A-Za-z_0-9]+)\s*(<|\()", name.replace("(anonymous namespace)::", "").replace("void ", ""))
    return m.group(1) if m else name[:60]


def avg(cs, n):
    return sum(cs[n]) / len(cs[n]) if n in cs else float("nan")


rows = sorted(((sum(cs.get("FETCH_SIZE", [0])) * 2 + sum(cs.get("WRITE_SIZE", [0])), k, cs) for k, cs in agg.items()), reverse=True)
fam = collections.defaultdict(lambda: [0.0, 0.0, 0, 0])
for _, k, cs in rows:
    f = fam[base(k)]
    f[0] += sum(cs.get("FETCH_SIZE", [])); f[1] += sum(cs.get("WRITE_SIZE", []))
    f[2] += len(cs.get("FETCH_SIZE", [])); f[3] += len(cs.get("WRITE_SIZE", []))
with open(out, "w") as o:
    o.write("# rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE TCC_HIT_sum TCC_MISS_sum) over `python bench.py --no-graph --steps 2 --warmup 1`\n")
    o.write("# per-launch averages; FETCH_SIZE/WRITE_SIZE in KB as reported.  gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide\n")
    o.write("# coalesced read -> hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md, HBM section)\n")
    o.write("## families (all template instantiations of a kernel name together)\n")
    o.write("kernel | launches | FETCH_SIZE_KB | WRITE_SIZE_KB | hbm_MB_corrected\n")
    fj = {}
    for k, (fs, ws, nf, nw) in sorted(fam.items(), key=lambda kv: -(2 * kv[1][0] + kv[1][1])):
        if not nf or not nw:
            continue
        f, w = fs / nf, ws / nw
        o.write(f"{k} | {nf} | {f:.1f} | {w:.1f} | {(2 * f + w) * 1024 / 1e6:.2f}\n")
        fj[k] = {"launches": nf, "hbm_bytes_per_launch": (2 * f + w) * 1024, "hbm_bytes_total": (2 * fs + ws) * 1024}
    o.write("## kernels\n")
    o.write("kernel | launches | FETCH_SIZE_KB | WRITE_SIZE_KB | hbm_MB_corrected | TCC_HIT | TCC_MISS\n")
    for _, k, cs in rows[:80]:
        f, w = avg(cs, "FETCH_SIZE"), avg(cs, "WRITE_SIZE")
        o.write(f"{k} | {len(cs.get('FETCH_SIZE', []))} | {f:.1f} | {w:.1f} | {(2 * f + w) * 1024 / 1e6:.2f} | {avg(cs, 'TCC_HIT_sum'):.0f} | {avg(cs, 'TCC_MISS_sum'):.0f}\n")
if fam_out:
    json.dump(fj, open(fam_out, "w"), indent=1)
print(open(out).read()[:3000])
