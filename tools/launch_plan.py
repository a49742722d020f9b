"""The launch plan of one inference step at the bench configuration as a table: layer -> kernel generation -> tile -> work decomposition ->
measured time in the step's sequence (events between consecutive launches, one batch at a time on one stream).
  python tools/launch_plan.py [--markdown]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, bench
E = importlib.import_module(bench.PKG + ".engine"); Wt = importlib.import_module(bench.PKG + ".weights"); L = importlib.import_module(bench.PKG + "._lib")
md = "--markdown" in sys.argv
state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=bench.CLS_BIAS, tame=True)
eng = E.Engine("resnet50", 1, 9, dtype="bf16"); eng.load_state(state)
x = bench.synth_images(torch, bench.BATCH, 1000, "cuda")
for _ in range(2): eng.detect(x)
reps = 5
per = eng.profile_ops(x, reps=reps)
plan = eng._plan(bench.BATCH, *bench.CANVAS)
ops = eng.active_ops(plan)
GEN = {1: "1 (128-row, register-staged)", 2: "2 (256-row LDS-DMA per tap)", 3: "3 (256-row shared halo)", 4: "4 (persistent 8-phase 3x3)",
       5: "5 (persistent 8-phase 1x1)", 6: "6 (narrow-N head output)"}
rows = []
eng._bind_stream()
for (kind, ms), op in zip(per, ops):
    ms /= reps
    name = op[2] if len(op) > 2 and isinstance(op[2], str) else kind
    if kind in ("conv", "dual"):
        eng._run_op(op, x); torch.cuda.synchronize()
        impl = L.lib.rtn_debug_last_conv_impl(eng.h.raw); tile = L.lib.rtn_debug_last_conv_tile(eng.h.raw); sk = L.lib.rtn_debug_last_conv_streamk(eng.h.raw)
        d = op[1]
        M = sum(d.g[i].Hout * d.g[i].Wout for i in range(d.ngroups)) * bench.BATCH
        K = d.KH * d.KW * d.Crun + (op[3].C if kind == "dual" else 0)
        shape = "M %d, N %d, K %d, %dx%d%s%s" % (M, d.N, K, d.KH, d.KW, " s2" if d.sy == 2 else "", ", 5 levels" if d.ngroups == 5 else "")
        tl = "%d x %d" % (tile >> 16, tile & 0xffff) if tile else "-"
        dec = ("stream-K on %d workgroups" % sk) if sk else ("K slices + finish" if (impl == 4 and d.workspace and M < 20000) else "whole tiles")
        rows.append((name, GEN.get(impl, str(impl)), tl, dec, shape, ms))
    elif kind == "bneck":
        m = op[3]
        rows.append((name, "fused bottleneck (rtn_bottleneck64_fwd)", "32-pixel strips per wave", "persistent, all filters in LDS", "M %d" % (m["B"] * m["H"] * m["W"]), ms))
    elif kind == "chain":
        m = op[3]
        rows.append((name, "fused seam (rtn_chain1x1_fwd)", "%d-pixel strips per wave" % (32 if m["mid"] == 128 else 16), "persistent, filters streamed through LDS in 64-channel chunks",
                     "M %d, %d -> %d -> %d" % (m["pixels"], m["mid"], 4 * m["mid"], m["mid"]), ms))
    elif kind == "stem":
        rows.append(("conv1 + ReLU + pool1 + res2a_branch2a", "fused stem (rtn_stem_conv_pool_branch2a)", "4 x 16 pooled pixels", "2 persistent workgroups per CU", "", ms))
    else:
        rows.append((name, kind, "", "", "", ms))
rows.append(("decode + threshold + NMS + top-k", "rtn_decode_filter_nms", "", "", "", per[-1][1] / reps))
tot = sum(r[-1] for r in rows)
if md:
    print("| layer | kernel | tile | decomposition | shape (batch 8) | µs |"); print("|---|---|---|---|---|---|")
    for r in rows: print("| %s | %s | %s | %s | %s | %.1f |" % (r[0], r[1], r[2], r[3], r[4], r[5] * 1e3))
    print("| **serial sum** | | | | | **%.0f** |" % (tot * 1e3))
else:
    for r in rows: print("%-38s %-34s %-10s %-28s %-44s %7.1f us" % (r[0], r[1], r[2], r[3], r[4], r[5] * 1e3))
    print("serial sum %.3f ms" % tot)
