"""Run the fused bottleneck ops of the bench plan N times (for rocprofv3 counter passes): python tools/run_bneck.py [N]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, bench
E = importlib.import_module(bench.PKG + ".engine"); Wt = importlib.import_module(bench.PKG + ".weights")
state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=bench.CLS_BIAS, tame=True)
eng = E.Engine("resnet50", 1, 9, dtype="bf16"); eng.load_state(state)
x = bench.synth_images(torch, bench.BATCH, 1000, "cuda")
eng.forward(x); torch.cuda.synchronize()
plan = eng._plan(bench.BATCH, *bench.CANVAS)
ops = [op for op in eng.active_ops(plan) if op[0] == "bneck"]
eng._bind_stream()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    for op in ops: eng._run_op(op, x)
torch.cuda.synchronize()
