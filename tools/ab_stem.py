"""Time of the fused stem launch (conv1 + ReLU + pool1 [+ res2a_branch2a]) at the bench size, events around 10 launches, median of 8 rounds."""
import importlib, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, bench
E = importlib.import_module(bench.PKG + ".engine"); Wt = importlib.import_module(bench.PKG + ".weights")
state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=bench.CLS_BIAS, tame=True)
eng = E.Engine("resnet50", 1, 9, dtype="bf16"); eng.load_state(state)
x = bench.synth_images(torch, bench.BATCH, 1000, "cuda")
for training in (False, True):
    eng.training = training
    eng.forward(x); torch.cuda.synchronize()
    plan = eng._plan(bench.BATCH, *bench.CANVAS)
    op = [o for o in eng.active_ops(plan) if o[0] == "stem"][0]
    ts = []
    for rnd in range(9):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        eng._bind_stream(); s.record()
        for _ in range(10): eng._run_op(op, x)
        e.record(); torch.cuda.synchronize()
        if rnd: ts.append(s.elapsed_time(e) / 10)
    print("stem launch, %s: %.4f ms (min %.4f)" % ("training (winning taps recorded)" if training else "inference", statistics.median(ts), min(ts)))
