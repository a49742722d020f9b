"""Same-process A/B of the fused bottleneck kernel's variants (RTN_BNECK_ROWPP) on the res2 blocks of the bench
plan, against the three separate launches they replace.  python tools/ab_bottleneck.py"""
import importlib, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, bench
E = importlib.import_module(bench.PKG + ".engine"); Wt = importlib.import_module(bench.PKG + ".weights")
state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=bench.CLS_BIAS, tame=True)
eng = E.Engine("resnet50", 1, 9, dtype="bf16"); eng.load_state(state)
x = bench.synth_images(torch, bench.BATCH, 1000, "cuda")
eng.detect(x); torch.cuda.synchronize()
plan = eng._plan(bench.BATCH, *bench.CANVAS)
fused = [op for op in eng.active_ops(plan) if op[0] == "bneck"]
plain = {op[2]: op for op in plan["ops"] if op[0] == "conv"}
variants = [{"RTN_BNECK_ROWPP": v} for v in ("1", "0")]     # the cross-strip software pipeline (the tap / store forms of round 4: profiles/r4_bottleneck_shifted_taps.txt)
if "--phase" in sys.argv:         # start delay per wave index (x 64 cycles)
    raise SystemExit("--phase: the per-wave start delay experiment (no effect, profiles/r2_v2_bottleneck_fused.txt) was removed from the kernel")
if "--ablate" in sys.argv:        # which stream bounds the kernel: drop one at a time (timing only, outputs are wrong)
    raise SystemExit("the RTN_BNECK_DBG ablation masks left the library in round 4: profiles/r2_v2_bottleneck_fused.txt holds the result")


def timed(fn, n=5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    eng._bind_stream(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for op in fused:
    m = op[3]; px = m["B"] * m["H"] * m["W"]
    by = 2.0 * px * (64 + 256 + 256 + (64 if m["tail"] else 0))
    base = op[2].split("+")[0]                      # resNx_branch2b
    names = [base, base.replace("branch2b", "branch2c")]
    if m["tail"]:
        blk = base[4]; names.append("res2%s_branch2a" % chr(ord(blk) + 1))
    times = {i: [] for i in range(len(variants) + 1)}
    for rnd in range(10):
        for i, v in enumerate(variants):
            os.environ.update(v)
            t = timed(lambda: eng._run_op(op, x))
            if rnd >= 2: times[i].append(t)
        t = timed(lambda: [eng._run_op(plain[n], x) for n in names])
        if rnd >= 2: times[len(variants)].append(t)
    for i, v in enumerate(variants):
        med = statistics.median(times[i])
        print("%-28s %s: median %.4f ms  min %.4f  (%.2f TB/s)" % (op[2], " ".join("%s=%s" % kv for kv in v.items()), med, min(times[i]), by / med / 1e9))
    med = statistics.median(times[len(variants)])
    print("%-28s separate launches %s: median %.4f ms" % (op[2], "+".join(n.split("_")[1] for n in names), med))
