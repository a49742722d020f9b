"""Which pairs of pool streams overlap?  Engine.in_flight=2 with its two slot streams re-created several times."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, bench
E = importlib.import_module(bench.PKG + ".engine"); Wt = importlib.import_module(bench.PKG + ".weights")
state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=bench.CLS_BIAS, tame=True)
x = bench.synth_images(torch, bench.BATCH, 1, "cuda")
eng = E.Engine("resnet50", 1, 9, dtype="bf16"); eng.load_state(state)
if os.environ.get("LANES_FIRST", "1") == "1":
    eng.detect(x); torch.cuda.synchronize()
def timed(n=40):
    for _ in range(4): eng.detect(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): eng.detect(x)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
eng.in_flight = 2
for k in range(10):
    eng.join(); torch.cuda.synchronize()
    eng._slots = []                     # forces two NEW streams from torch's pool
    t = timed()
    print("slot set %d: %.3f ms/step (streams %s)" % (k, t, [hex(sl["stream"].cuda_stream)[-5:] for sl in eng._slots]), flush=True)
