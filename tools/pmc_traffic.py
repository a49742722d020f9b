"""HBM traffic per step of the bench's conv launches from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE need separate
passes: MI355X_MICROARCH.md, rocprofv3 PMC slots):

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary
  python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> 7.125 profiles/rN_pmc_traffic.json [bench stdout]

The third argument is the number of batch-8 forward passes the process ran (warmup + steps + 3 per-op profiling passes + 1/8 for the
batch-1 bias calibration).  FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled (gfx950 tallies the 128-byte requests of
16 B/lane loads at 64 B: MI355X_MICROARCH.md, HBM section).  bench.py quotes the result only while the launch count per step
matches the plan it runs: pass the stdout of one of the two bench runs as the last argument and its
roofline.launches_per_step (conv ops of the plan, which is what bench.py counts) is recorded as bench_launches_per_step."""
import csv
import json
import re
import subprocess
import sys
from collections import defaultdict


def per_kernel(path, counter):
    tot, calls, seen = defaultdict(float), defaultdict(int), set()
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            name = re.sub(r"\(.*", "", name).replace("void ", "")
            tot[name] += float(r["Counter_Value"])
            key = (r["Dispatch_Id"], name)
            if key not in seen:
                seen.add(key)
                calls[name] += 1
    return tot, calls


def main():
    fetch_csv, write_csv, out = sys.argv[1], sys.argv[2], sys.argv[4]
    f, fc = per_kernel(fetch_csv, "FETCH_SIZE")
    w, wc = per_kernel(write_csv, "WRITE_SIZE")
    if sys.argv[3] == "auto":        # forward passes of the process = dispatches of the stem kernel (exactly one per pass): a constant here
        passes = float(max(n for k, n in fc.items() if k.startswith("stem_fused")))        # went stale when bench.py gained a leg (r4_v2 read 12 % high)
    else:
        passes = float(sys.argv[3])
    kernels = {}
    for name in sorted(set(f) | set(w)):
        kernels[name] = {"launches_per_step": round(max(fc.get(name, 0), wc.get(name, 0)) / passes, 3),
                         "fetch_bytes_per_step": int(2 * 1024 * f.get(name, 0.0) / passes),
                         "write_bytes_per_step": int(1024 * w.get(name, 0.0) / passes)}
    conv = {k: v for k, v in kernels.items() if k.startswith(("conv_", "stem_fused", "bottleneck64", "chain1x1"))}
    fb = sum(v["fetch_bytes_per_step"] for v in conv.values())
    wb = sum(v["write_bytes_per_step"] for v in conv.values())
    n = sum(v["launches_per_step"] for v in conv.values())
    bench_launches = None
    if len(sys.argv) > 5:
        for line in open(sys.argv[5]):
            line = line.strip()
            if line.startswith("{") and '"roofline"' in line:
                bench_launches = json.loads(line)["roofline"]["launches_per_step"]
    import os
    commit = os.environ.get("RTN_GIT_COMMIT")             # the GPU box has no .git: the collecting script passes the commit
    if not commit:
        try:
            commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True).strip()
        except Exception:
            commit = "unknown"
    json.dump({"_how": __doc__, "git_commit": commit, "batch8_passes": passes, "bench_launches_per_step": bench_launches,
               "kernels": kernels,
               "conv_total": {"launches_per_step": n, "fetch_bytes_per_step": fb, "write_bytes_per_step": wb,
                              "hbm_bytes_per_step": fb + wb, "hbm_bytes_per_launch": int((fb + wb) / max(n, 1e-9))}},
              open(out, "w"), indent=1)
    print("conv launches/step %.2f, HBM bytes/step %.3f GB (fetch %.3f, write %.3f)" % (n, (fb + wb) / 1e9, fb / 1e9, wb / 1e9))


if __name__ == "__main__":
    main()
