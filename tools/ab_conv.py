"""Same-process A/B of conv kernel variants on one layer shape (interleaved rounds, median and min)."""
import ctypes as C, importlib, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, bench
E = importlib.import_module(bench.PKG + ".engine"); Wt = importlib.import_module(bench.PKG + ".weights"); L = importlib.import_module(bench.PKG + "._lib")
layers = sys.argv[1].split(",") if len(sys.argv) > 1 else ["pyramid_regression_1"]
variants = [dict(kv.split("=") for kv in v.split("+")) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["RTN_CONV_IMPL=1", "RTN_CONV_IMPL=2"])]
state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=bench.CLS_BIAS, tame=True)
eng = E.Engine("resnet50", 1, 9, dtype="bf16"); eng.load_state(state)
x = bench.synth_images(torch, bench.BATCH, 1000, "cuda")
eng.detect(x); torch.cuda.synchronize()
plan = eng._plan(bench.BATCH, *bench.CANVAS)
ops = {op[2]: op for op in plan["ops"] if op[0] == "conv"}
ops.update({op[2]: op for op in eng.active_ops(plan) if op[0] == "dual"})        # e.g. res2a_branch2c+1 (folded shortcut)
if layers == ["all"]:
    layers = [n for n in ops if n != "conv1"]
KNOBS = sorted({k for v in variants for k in v})
for name in layers:
    op = ops[name]; fl = bench.conv_flops(op[1], bench.BATCH)
    if op[0] == "dual": eng._dual_weights()
    times = {i: [] for i in range(len(variants))}
    for rnd in range(12):
        for i, v in enumerate(variants):
            for k in KNOBS: os.environ.pop(k, None)
            os.environ.update(v)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            eng._bind_stream(); s.record()
            for _ in range(5): eng._run_op(op, x)
            e.record(); torch.cuda.synchronize()
            if rnd >= 2: times[i].append(s.elapsed_time(e) / 5)
    for i, v in enumerate(variants):
        med, mn = statistics.median(times[i]), min(times[i])
        print("%-26s %-36s median %.4f ms (%.0f TF/s)  min %.4f ms (%.0f TF/s)" % (name, "+".join("%s=%s" % kv for kv in v.items()), med, fl / med / 1e9, mn, fl / mn / 1e9))
