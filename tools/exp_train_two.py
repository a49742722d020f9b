"""Experiment: the batch-16 training step as two concurrent micro-batches of 8 (two engines + trainers on two HIP streams) against one
trainer on the whole batch.  forward + loss + backward only (the optimizer step is shared work)."""
import ctypes as C, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
E = importlib.import_module(bench.PKG + ".engine"); Wt = importlib.import_module(bench.PKG + ".weights"); T = importlib.import_module(bench.PKG + ".trainer")
state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=-2.0, tame=True)
H, W = bench.CANVAS
cfg, N = E.make_anchor_cfg(bench.CANVAS)
def mk(B, lanes):
    eng = E.Engine("resnet50", 1, 9, dtype="bf16"); eng.load_state(state)
    if not lanes: eng.two_streams = False
    tr = T.Trainer(eng, lr=1e-4, clipnorm=0.001)
    if not lanes: tr.wgrad_lane = False
    x = bench.synth_images(torch, B, 7, "cuda")
    reg_t = torch.zeros(B, N, 5, device="cuda"); lab_t = torch.zeros(B, N, 2, device="cuda")
    lab_t[:, ::211, 0] = 1; lab_t[:, ::211, 1] = 1; reg_t[:, ::211, 4] = 1
    return tr, x, reg_t, lab_t
def timed(fn, n=8):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
tr16, x16, r16, l16 = mk(16, True)
print("one trainer, batch 16, lanes on:        %.2f ms per 16 images" % timed(lambda: tr16.forward_backward(x16, r16, l16)), flush=True)
del tr16
torch.cuda.empty_cache()
for lanes in (False, True):
    pair = [mk(8, lanes) for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    def both():
        for (tr, x, r, l), st in zip(pair, streams):
            with torch.cuda.stream(st):
                tr.forward_backward(x, r, l)
    print("two trainers, batch 8 each, lanes %s: %.2f ms per 16 images" % ("on " if lanes else "off", timed(both)), flush=True)
    del pair
    torch.cuda.empty_cache()
