"""The HBM-bound kernels of the path against the achievable HBM rate (6.3 TB/s): algorithmic bytes per launch (every tensor
read once, written once) / average launch duration from `rocprofv3 --kernel-trace --stats` of the bench commands.
  python tools/hbm_table.py <train_kernel_stats_one_stream.csv> <bench_kernel_stats_one_stream.csv>
Both traces must be the ONE-STREAM ones (RTN_WGRAD_LANE=0 RTN_TWO_STREAMS=0 / --in-flight 1): with lanes a short kernel's recorded duration
includes the time it shares the device with its neighbours (the finish passes read 5-8 us alone and 100 us beside a weight gradient)."""
import csv, ctypes as C, importlib, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, bench
E = importlib.import_module(bench.PKG + ".engine"); Wt = importlib.import_module(bench.PKG + ".weights"); T = importlib.import_module(bench.PKG + ".trainer")
L = importlib.import_module(bench.PKG + "._lib")


def stats(path):
    out = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            name = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).replace("void ", "")
            name = re.sub(r"\(.*", "", name)
            c, t = out.get(name, (0, 0.0))
            calls, total = int(r["Calls"]), float(r["TotalDurationNs"])
            if name.startswith("pack_dgrad") and calls > 1:      # without the process's first dispatch (profiles/r4_first_repack.txt)
                total = (total - float(r["MaxNs"])) * calls / (calls - 1)
            out[name] = (c + calls, t + total)
    return out


tr_stats, inf_stats = stats(sys.argv[1]), stats(sys.argv[2])
state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=-2.0, tame=True)
eng = E.Engine("resnet50", 1, 9, dtype="bf16"); eng.load_state(state)
tr = T.Trainer(eng, lr=1e-4, clipnorm=0.001)
B, (H, W) = bench.TRAIN_BATCH, bench.CANVAS
bp = tr._bplan(B, H, W)
N = bp["plan"]["N"]
rows = B * N
H1, W1 = (H + 1) // 2, (W + 1) // 2
Hp, Wp = (H1 + 1) // 2, (W1 + 1) // 2
xi = [op for op in bp["plan"]["ops"] if op[0] == "pack"][0][2]
steps_t = tr_stats["loss_fwd_kernel"][0]                      # one call per training step
steps_i = inf_stats["detect_candidates_kernel"][0]            # one call per inference step
finish_bytes = 0.0
for b in bp["bops"]:
    if b[0] == "wgrad":
        d = b[1]
        NK = d.N * d.KH * d.KW * d.Crun
        tiles = sum((d.g[i].Hout * d.g[i].Wout * B + 63) // 64 for i in range(d.ngroups))
        table = ((tiles * 64 * 16 + 255) // 256) * 256
        slabs = max(0, int(L.lib.rtn_conv2d_wgrad_workspace_bytes(C.byref(d))) - table)
        finish_bytes += slabs + 2 * NK * 4
padcast = sum(b[3] * (b[4] * 4 + b[5] * 2) for b in bp["bops"] if b[0] == "padcast")
packd = 2.0 * sum(w.numel() for w in tr.wd.values()) * 2
entries = [   # (kernel, where, bytes per step, launches per step or None = from the stats)
    ("stem_pack_kernel<0, 0>", "train", B * (H * W * 3 * 2 + xi["Hp"] * xi["Wp"] * 4 * 2)),
    ("anchor_targets_kernel", "train", rows * (5 + 2) * 4),
    ("loss_fwd_kernel", "train", rows * (2 + 5 + 1 + 4) * 4),
    ("loss_bwd_kernel", "train", rows * ((2 + 5 + 1 + 4) * 4 + 5 * 4)),
    ("sumsq_kernel", "train", (tr.NW + tr.NB) * 4 * 2),
    ("adam_kernel<2>", "train", tr.NW * (6 * 4 + 3 * 4 + 2)),
    ("adam_kernel<4>", "train", tr.NB * (6 * 4 + 3 * 4 + 4)),
    ("maxpool_bwd_idx_kernel<2>", "train", B * (Hp * Wp * 64 * (2 + 1 + 2) + H1 * W1 * 64 * 2)),
    ("pad_cast_rows8_kernel", "train", padcast),
    ("wgrad_finish_kernel (all instances)", "train", finish_bytes),
    ("pack_dgrad_multi_kernel<unsigned short>", "train", packd),
    ("stem_pack_kernel<0, 0>", "infer", bench.BATCH * (H * W * 3 * 2 + xi["Hp"] * xi["Wp"] * 4 * 2)),
    ("detect_candidates_kernel", "infer", bench.BATCH * N * 4),
    ("relu_kernel<2>", "infer", None),
    ("nms_sort_kernel", "infer", None), ("nms_mask_kernel", "infer", None), ("nms_kernel", "infer", None), ("merge_topk_kernel", "infer", None),
]
print("%-42s %-6s %14s %10s %10s %8s" % ("kernel", "step", "bytes / step", "us / step", "GB/s", "of 6.3T"))
for name, where, by in entries:
    st, steps = (tr_stats, steps_t) if where == "train" else (inf_stats, steps_i)
    if name.startswith("wgrad_finish"):
        c = sum(v[0] for k, v in st.items() if k.startswith("wgrad_finish_kernel")); t = sum(v[1] for k, v in st.items() if k.startswith("wgrad_finish_kernel"))
    else:
        c, t = st.get(name, (0, 0.0))
    if not c:
        print("%-42s %-6s (not in this trace)" % (name, where)); continue
    us = t / steps / 1e3
    if by is None:
        print("%-42s %-6s %14s %10.1f %10s %8s   (%.1f launches / step; candidates only: latency-bound)" % (name, where, "-", us, "-", "-", c / steps))
    else:
        gbs = by / (us * 1e-6) / 1e9
        print("%-42s %-6s %14.0f %10.1f %10.0f %8.2f   (%.1f launches / step)" % (name, where, by, us, gbs, gbs / 6300.0, c / steps))
