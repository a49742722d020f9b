"""Run one conv layer of the bench plan N times (for rocprofv3 --pmc runs)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, bench
E = importlib.import_module(bench.PKG + ".engine"); Wt = importlib.import_module(bench.PKG + ".weights")
names = sys.argv[1].split(","); n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=bench.CLS_BIAS, tame=True)
eng = E.Engine("resnet50", 1, 9, dtype="bf16"); eng.load_state(state)
x = bench.synth_images(torch, bench.BATCH, 1000, "cuda")
eng.forward(x); torch.cuda.synchronize()
ops = {op[2]: op for op in eng._plan(bench.BATCH, *bench.CANVAS)["ops"] if op[0] == "conv"}
eng._bind_stream()
for nm in names:
    for _ in range(n): eng._run_op(ops[nm], x)
torch.cuda.synchronize()
