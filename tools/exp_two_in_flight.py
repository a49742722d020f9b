"""Experiment: two independent batch-8 steps in flight (two engines = two buffer sets, each with its own lanes)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, bench
E = importlib.import_module(bench.PKG + ".engine"); Wt = importlib.import_module(bench.PKG + ".weights")
state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=bench.CLS_BIAS, tame=True)
dev = torch.device("cuda", 0)
engs = [E.Engine("resnet50", 1, 9, dtype="bf16") for _ in range(2)]
for e in engs: e.load_state(state)
xs = [bench.synth_images(torch, bench.BATCH, 1 + i, "cuda") for i in range(2)]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
def one(n):
    for _ in range(3): engs[0].detect(xs[0])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): engs[0].detect(xs[0])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def two(n):
    for i in range(4):
        with torch.cuda.stream(streams[i & 1]): engs[i & 1].detect(xs[i & 1])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        with torch.cuda.stream(streams[i & 1]): engs[i & 1].detect(xs[i & 1])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(2):
    print("one in flight: %.3f ms/step   two in flight: %.3f ms/step" % (one(40), two(40)), flush=True)
