"""Warm vs cold: one conv layer timed (a) back to back on the same buffers, (b) with the caches flushed before every launch (a 1 GiB
fill), (c) right after its producer layer, as in the step.  python tools/exp_cold.py LAYERS [VARIANTS]"""
import importlib, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, bench
E = importlib.import_module(bench.PKG + ".engine"); Wt = importlib.import_module(bench.PKG + ".weights")
layers = sys.argv[1].split(",")
variants = [dict(kv.split("=") for kv in v.split("+") if kv) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else [""])]
state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=bench.CLS_BIAS, tame=True)
eng = E.Engine("resnet50", 1, 9, dtype="bf16"); eng.load_state(state)
x = bench.synth_images(torch, bench.BATCH, 1000, "cuda")
eng.detect(x); torch.cuda.synchronize()
plan = eng._plan(bench.BATCH, *bench.CANVAS)
act = eng.active_ops(plan)
names = [op[2] if len(op) > 2 and isinstance(op[2], str) else op[0] for op in act]
flush = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
KNOBS = sorted({k for v in variants for k in v})
def timed(fn, pre=None, n=12):
    ts = []
    for _ in range(n):
        if pre: pre()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        eng._bind_stream(); s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return statistics.median(ts[2:])
for name in layers:
    i = names.index(name); op = act[i]
    for v in variants:
        for k in KNOBS: os.environ.pop(k, None)
        os.environ.update(v)
        warm = timed(lambda: eng._run_op(op, x))
        cold = timed(lambda: eng._run_op(op, x), pre=lambda: flush.fill_(1))
        def after_producer():
            pass
        seq = timed(lambda: eng._run_op(op, x), pre=lambda: [eng._run_op(act[j], x) for j in range(max(0, i - 3), i)])
        print("%-24s %-40s warm %.4f  flushed %.4f  after its 3 producers %.4f ms" % (name, "+".join("%s=%s" % kv for kv in v.items()) or "default", warm, cold, seq))
