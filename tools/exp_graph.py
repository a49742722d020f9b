"""Experiment: capture one detect() step (all lanes) into a HIP graph and replay it."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, bench
E = importlib.import_module(bench.PKG + ".engine"); Wt = importlib.import_module(bench.PKG + ".weights")
state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=bench.CLS_BIAS, tame=True)
eng = E.Engine("resnet50", 1, 9, dtype="bf16"); eng.load_state(state)
x = bench.synth_images(torch, bench.BATCH, 1, "cuda")
for _ in range(3): out = eng.detect(x)
torch.cuda.synchronize()
ref = [t.clone() for t in out]
def timeit(fn, n=30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager   %.3f ms/step" % timeit(lambda: eng.detect(x)))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out_g = eng.detect(x)
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print("graph outputs equal eager:", all(torch.equal(a, b) for a, b in zip(out_g, ref)))
print("graph   %.3f ms/step" % timeit(g.replay))
print("eager   %.3f ms/step" % timeit(lambda: eng.detect(x)))
print("graph   %.3f ms/step" % timeit(g.replay))
