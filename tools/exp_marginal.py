"""Marginal cost of each group of launches with N batches in flight: the step timed with that group's launches skipped (wrong results).
  python tools/exp_marginal.py [in_flight]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, bench
E = importlib.import_module(bench.PKG + ".engine"); Wt = importlib.import_module(bench.PKG + ".weights")
n_if = int(sys.argv[1]) if len(sys.argv) > 1 else 2
state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=bench.CLS_BIAS, tame=True)
eng = E.Engine("resnet50", 1, 9, dtype="bf16"); eng.load_state(state)
x = bench.synth_images(torch, bench.BATCH, 1, "cuda")
eng.in_flight = n_if
skip = [None]
orig = eng._run_op
def name_of(op):
    return op[2] if len(op) > 2 and isinstance(op[2], str) else op[0]
def run_op(op, images):
    if skip[0] is not None and skip[0](name_of(op)):
        return
    orig(op, images)
eng._run_op = run_op
def timed(n=40):
    for _ in range(6): eng.detect(x)
    eng.join(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): eng.detect(x)
    eng.join(); torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
groups = [("nothing", None),
          ("head towers", lambda n: n.startswith(("pyramid_regression_", "pyramid_classification_"))),
          ("head outputs", lambda n: n in ("pyramid_regression", "pyramid_classification")),
          ("stem + pack", lambda n: n in ("stem", "pack")),
          ("res2", lambda n: n.startswith("res2")),
          ("res3", lambda n: n.startswith("res3")),
          ("res4", lambda n: n.startswith("res4")),
          ("res5", lambda n: n.startswith("res5")),
          ("FPN (C*_reduced, P3-P7, relu)", lambda n: n.startswith(("C3", "C4", "C5", "P3", "P4", "P5", "P6", "P7", "relu"))),
          ("res3 1x1 layers", lambda n: n.startswith("res3") and ("branch2a" in n or "branch2c" in n)),
          ("res4 1x1 layers", lambda n: n.startswith("res4") and ("branch2a" in n or "branch2c" in n)),
          ("res4 3x3 layers", lambda n: n.startswith("res4") and n.endswith("branch2b"))]
base = None
for tag, fn in groups:
    skip[0] = fn
    t = timed()
    if base is None: base = t
    print("%d in flight, without %-32s %.3f ms/step  (marginal %.3f ms)" % (n_if, tag + ":", t, base - t), flush=True)
