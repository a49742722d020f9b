"""MFMA-pipe busy share and wave-cycle shares per kernel from two rocprofv3 PMC passes over bench.py:

  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d A -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d B -- python3 bench.py ...
  python tools/pmc_mfma_util.py <A counter_collection.csv> <B counter_collection.csv> out.json

mfma_util_pct = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs) * 100 (the CSV sums GRBM_GUI_ACTIVE over the 8 XCDs);
conv_halo8_kernel additionally per dispatch (median over the full-chip dispatches: grid = 256 workgroups)."""
import csv, json, re, statistics, sys
from collections import defaultdict


def load(path):
    per, disp = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(float))
    grid = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            name = re.sub(r"\(.*", "", name).replace("void ", "")
            per[name][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[(name, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
            grid[(name, r["Dispatch_Id"])] = int(r.get("Grid_Size", 0) or 0)
    return per, disp, grid


a, da, ga = load(sys.argv[1])
b, _, _ = load(sys.argv[2])
out = {"_how": __doc__, "kernels": {}}
for name in sorted(a):
    busy, act = a[name].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), a[name].get("GRBM_GUI_ACTIVE", 0.0)
    if act <= 0 or busy <= 0:
        continue
    e = {"dispatches": sum(1 for k in da if k[0] == name), "mfma_util_pct": round(100.0 * busy / (act / 8 * 1024), 1)}
    wc = b.get(name, {}).get("SQ_WAVE_CYCLES", 0.0)
    if wc > 0:
        e.update(parked_pct=round(100 * b[name]["SQ_WAIT_ANY"] / wc, 1), issue_stalled_pct=round(100 * b[name]["SQ_WAIT_INST_ANY"] / wc, 1),
                 issuing_pct=round(100 * b[name]["SQ_ACTIVE_INST_ANY"] / wc, 1))
    full = [100.0 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8 * 1024) for k, v in da.items()
            if k[0] == name and ga.get(k, 0) == 256 * 512 and v.get("GRBM_GUI_ACTIVE", 0) > 0]
    if full and name.startswith(("conv_halo8", "conv_gemm8", "conv_halon")):
        e["full_chip_dispatches"] = len(full)
        e["full_chip_mfma_util_pct_median"] = round(statistics.median(full), 1)
        e["full_chip_mfma_util_pct_min_max"] = [round(min(full), 1), round(max(full), 1)]
    out["kernels"][name] = e
json.dump(out, open(sys.argv[3], "w"), indent=1)
top = sorted(out["kernels"].items(), key=lambda kv: -kv[1]["mfma_util_pct"])[:12]
for k, v in top:
    print("%-64s %s" % (k[:64], v))
