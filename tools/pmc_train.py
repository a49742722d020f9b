"""HBM traffic and MFMA-pipe busy share of the TRAINING step from rocprofv3 PMC passes over `bench.py --mode train`:

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d A -- python3 bench.py --mode train --steps 3 --warmup 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d B -- python3 bench.py --mode train --steps 3 --warmup 1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d C -- python3 bench.py --mode train --steps 3 --warmup 1
  python tools/pmc_train.py <A csv> <B csv> <C csv> <steps run = warmup + steps> <batch> <git commit> out_traffic.json out_mfma.json

FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled (gfx950 tallies the 128-byte requests of 16 B/lane loads at 64 B:
MI355X_MICROARCH.md, HBM section).  Every kernel the process ran is counted (torch's few fill / copy kernels included) and
divided by the number of steps; per-kernel rows are kept so that the step's total can be re-derived."""
import csv, json, re, sys
from collections import defaultdict


def load(path):
    tot, calls = defaultdict(lambda: defaultdict(float)), defaultdict(set)
    with open(path) as f:
        for r in csv.DictReader(f):
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            name = re.sub(r"\(.*", "", name).replace("void ", "")
            tot[name][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[name].add(r["Dispatch_Id"])
    return tot, calls


fa, fb, fc, steps, batch, commit, out_t, out_m = sys.argv[1:9]
steps = float(steps)
f, fcalls = load(fa)
w, _ = load(fb)
m, mcalls = load(fc)
kern = {}
for name in sorted(set(f) | set(w)):
    fe, wr = f.get(name, {}).get("FETCH_SIZE", 0.0) * 1024 * 2, w.get(name, {}).get("WRITE_SIZE", 0.0) * 1024
    kern[name] = {"launches_per_step": len(fcalls.get(name, ())) / steps, "fetch_bytes_per_step": fe / steps, "write_bytes_per_step": wr / steps}
total = sum(k["fetch_bytes_per_step"] + k["write_bytes_per_step"] for k in kern.values())
json.dump({"_how": __doc__, "git_commit": commit, "steps": steps, "batch": int(batch), "hbm_bytes_per_step": total,
           "fetch_bytes_per_step": sum(k["fetch_bytes_per_step"] for k in kern.values()),
           "write_bytes_per_step": sum(k["write_bytes_per_step"] for k in kern.values()),
           "kernels": dict(sorted(kern.items(), key=lambda kv: -(kv[1]["fetch_bytes_per_step"] + kv[1]["write_bytes_per_step"])))}, open(out_t, "w"), indent=1)
util = {}
for name in sorted(m):
    busy, act = m[name].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), m[name].get("GRBM_GUI_ACTIVE", 0.0)
    if busy > 0 and act > 0:
        util[name] = {"dispatches": len(mcalls[name]), "mfma_util_pct": round(100.0 * busy / (act / 8 * 1024), 1),
                      "gui_active_cycles_per_dispatch": round(act / 8 / len(mcalls[name]))}
json.dump({"_how": __doc__, "git_commit": commit,
           "kernels": dict(sorted(util.items(), key=lambda kv: -kv[1]["mfma_util_pct"]))}, open(out_m, "w"), indent=1)
print("training step: %.2f GB of HBM traffic per step (fetch %.2f + write %.2f)" % (total / 1e9, sum(k["fetch_bytes_per_step"] for k in kern.values()) / 1e9,
                                                                                   sum(k["write_bytes_per_step"] for k in kern.values()) / 1e9))
for k, v in list(sorted(util.items(), key=lambda kv: -kv[1]["dispatches"] * kv[1]["gui_active_cycles_per_dispatch"]))[:14]:
    print("%-64s %s" % (k[:64], v))
