"""Host time to enqueue one detect() step vs its device time (is the launch loop ahead of the GPU?)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, bench
E = importlib.import_module(bench.PKG + ".engine"); Wt = importlib.import_module(bench.PKG + ".weights")
state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=bench.CLS_BIAS, tame=True)
eng = E.Engine("resnet50", 1, 9, dtype="bf16"); eng.load_state(state)
x = bench.synth_images(torch, bench.BATCH, 1, "cuda")
for _ in range(3): eng.detect(x)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n): eng.detect(x)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.3f ms/step, total %.3f ms/step (device-bound if enqueue << total)" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
