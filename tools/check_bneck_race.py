"""Run one fused bottleneck op repeatedly on fixed inputs; report where outputs differ between runs (pixel within strip, channel)."""
import importlib, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
E = importlib.import_module(bench.PKG + ".engine"); Wt = importlib.import_module(bench.PKG + ".weights")
state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=bench.CLS_BIAS, tame=True)
eng = E.Engine("resnet50", 1, 9, dtype="bf16"); eng.load_state(state)
x = bench.synth_images(torch, bench.BATCH, 1000, "cuda")
eng.forward(x); torch.cuda.synchronize()
plan = eng._plan(bench.BATCH, *bench.CANVAS)
ops = [op for op in eng.active_ops(plan) if op[0] == "bneck"]
eng._bind_stream()
for op in ops:
    outs = op[3]["ys"]
    ref = None
    print(op[2])
    for rep in range(6):
        for t in outs: t.fill_(-9.0)
        eng._run_op(op, x); torch.cuda.synchronize()
        cur = [t.clone() for t in outs]
        if ref is None:
            ref = cur; continue
        for ti, (a, b) in enumerate(zip(ref, cur)):
            d = (a.float() - b.float()).abs().flatten(0, 2)          # [pixels][channels]
            bad = (d > 0).nonzero()
            if len(bad) == 0:
                print("  run %d tensor %d: identical" % (rep, ti)); continue
            px, ch = bad[:, 0].cpu().numpy(), bad[:, 1].cpu().numpy()
            print("  run %d tensor %d: %d elements differ over %d pixels; pixel%%32 histogram %s; channel//8 set size %d; example pixels %s; chans of first pixel %s" %
                  (rep, ti, len(bad), len(set(px.tolist())), dict(collections.Counter((px % 32).tolist())), len(set((ch // 8).tolist())),
                   sorted(set(px.tolist()))[:6], sorted(ch[px == px[0]].tolist())[:16]))
