"""Stations of conv_gemm8_kernel in time (needs a -DRTN_G8_STAMP build passed as RTN_LIB_PATH): per workgroup, microseconds since the
first workgroup started: 0 entry, 1 prologue done, 2 last K loop done, 3 closing barrier, 4 hand-off done, 5 epilogue issued, 6 drained.
  RTN_LIB_PATH=/path/librtn_stamp.so python tools/g8_stamps.py LAYER [ENV=VAL ...]"""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
for kv in sys.argv[2:]:
    k, v = kv.split("="); os.environ[k] = v
import numpy as np, torch, bench
E = importlib.import_module(bench.PKG + ".engine"); Wt = importlib.import_module(bench.PKG + ".weights"); L = importlib.import_module(bench.PKG + "._lib")
state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=bench.CLS_BIAS, tame=True)
eng = E.Engine("resnet50", 1, 9, dtype="bf16"); eng.load_state(state)
x = bench.synth_images(torch, bench.BATCH, 1000, "cuda")
eng.detect(x); torch.cuda.synchronize()
plan = eng._plan(bench.BATCH, *bench.CANVAS)
ops = {op[2]: op for op in eng.active_ops(plan) if op[0] in ("conv", "dual")}
op = ops[sys.argv[1]]
f = C.CDLL(L.LIB_PATH).rtn_debug_g8_stamps
for rep in range(3):
    eng._bind_stream(); eng._run_op(op, x); torch.cuda.synchronize()
buf = (C.c_ulonglong * 8192)()
assert f(buf) == 0
a = np.array(buf, dtype=np.uint64).reshape(1024, 8).astype(np.int64)
a = a[a[:, 0] > 0]
a = a[a[:, 0] >= a[:, 0].max() - 20000]          # the workgroups of the LAST launch (entries within 200 us of the latest)
t0 = a[:, 0].min()
us = (a[:, :7] - t0) / 100.0
names = ["entry", "prologue", "k-loop", "barrier", "hand-off", "epilogue", "drained"]
print("%s  %s  workgroups %d (stream-K grid %d)" % (sys.argv[1], " ".join(sys.argv[2:]), len(a), L.lib.rtn_debug_last_conv_streamk(eng.h.raw)))
for i, nm in enumerate(names):
    print("  %-9s min %6.2f  median %6.2f  max %6.2f us" % (nm, us[:, i].min(), np.median(us[:, i]), us[:, i].max()))
kl = (a[:, 6] - a[:, 0]) / 100.0
print("  workgroup life: median %.2f us at %.2f GHz (shader clocks / 100 MHz clock)" % (np.median(kl), np.median(a[:, 7] / np.maximum(kl, 1e-3)) / 1e3))
