"""Matrix-core busy fraction per kernel family from one `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE`
pass over bench.py (north_star asks for "rocprof HBM GB/s and MFMA-busy against gfx950 peak").

    python tools/pmc_mfma_sum.py '<glob of *counter_collection.csv>' <out.txt> [<families.json>]

mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024): the counter adds up, over the chip's 1024 SIMDs (256 CUs x 4), the
cycles each matrix pipe was occupied (32 per v_mfma_f32_32x32x16_bf16, 16 per v_mfma_f32_16x16x32_bf16: MI355X_MICROARCH.md, cycle
constants); rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (same file, DVFS give-back), so GRBM_GUI_ACTIVE / 8 is the
launch's duration in shader cycles.  1.0 = every matrix pipe issuing back to back for the whole launch = the dense peak AT THE CLOCK THE
CHIP HELD (eff_clock_GHz = cycles / duration; reads high on launches well under 0.3 ms, see that file)."""
import collections
import csv
import glob
import json
import re
import sys

pat, out = sys.argv[1], sys.argv[2]
fam_out = sys.argv[3] if len(sys.argv) > 3 else None
csv.field_size_limit(1 << 30)


def base(name):
    m = re.search(r"([A-Za-z_0-9]+)\s*(<|\()", name.replace("(anonymous namespace)::", "").replace("void ", ""))
    return m.group(1) if m else name[:60]


disp = collections.defaultdict(dict)        # (file, dispatch id) -> counters + name + duration
for f in glob.glob(pat):
    for r in csv.DictReader(open(f)):
        d = disp[(f, r["Dispatch_Id"])]
        d["name"] = r["Kernel_Name"]
        d[r["Counter_Name"]] = float(r["Counter_Value"])
        d["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
fam = collections.defaultdict(lambda: collections.defaultdict(float))
per = collections.defaultdict(lambda: collections.defaultdict(float))
for d in disp.values():
    if "GRBM_GUI_ACTIVE" not in d:
        continue
    for key, tgt in ((base(d["name"]), fam), (d["name"][:160], per)):
        a = tgt[key]
        a["n"] += 1
        for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_MFMA", "GRBM_GUI_ACTIVE", "ns"):
            a[c] += d.get(c, 0.0)


def row(k, a):
    cyc = a["GRBM_GUI_ACTIVE"] / 8.0
    busy = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0) if cyc else 0.0
    return (f"{k} | {int(a['n'])} | {a['ns'] / a['n'] / 1e3:.1f} | {busy:.3f} | {a['SQ_INSTS_MFMA'] / a['n']:.0f} | "
            f"{a['SQ_VALU_MFMA_BUSY_CYCLES'] / max(a['SQ_INSTS_MFMA'], 1):.1f} | {cyc / max(a['ns'], 1):.2f}"), busy


fj = {}
with open(out, "w") as o:
    o.write("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE over `python3 bench.py --no-graph --steps 2 --warmup 1`\n")
    o.write("# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs); duration under the counter pass (serialised dispatches)\n")
    o.write("## families\nkernel | launches | avg_us | mfma_busy | MFMA insts per launch | busy cycles per MFMA | eff_clock_GHz\n")
    for k, a in sorted(fam.items(), key=lambda kv: -kv[1]["SQ_VALU_MFMA_BUSY_CYCLES"]):
        if a["SQ_INSTS_MFMA"] == 0 and not k.startswith(("conv", "gemm", "tattn", "sattn")):
            continue
        line, busy = row(k, a)
        o.write(line + "\n")
        fj[k] = {"launches": int(a["n"]), "mfma_busy": busy, "avg_us": a["ns"] / a["n"] / 1e3,
                 "busy_cycles": a["SQ_VALU_MFMA_BUSY_CYCLES"], "gui_active": a["GRBM_GUI_ACTIVE"]}
    o.write("## kernels (template instantiations)\nkernel | launches | avg_us | mfma_busy | MFMA insts per launch | busy cycles per MFMA | eff_clock_GHz\n")
    for k, a in sorted(per.items(), key=lambda kv: -kv[1]["SQ_VALU_MFMA_BUSY_CYCLES"])[:60]:
        if a["SQ_INSTS_MFMA"] == 0:
            continue
        o.write(row(k, a)[0] + "\n")
if fam_out:
    json.dump(fj, open(fam_out, "w"), indent=1)
print(open(out).read()[:2500])
