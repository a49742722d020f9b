"""Per-op device time of one forward pass at the bench configuration (events on the launch stream)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
PKG = bench.PKG
E = importlib.import_module(PKG + ".engine"); Wt = importlib.import_module(PKG + ".weights")
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else bench.BATCH
state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=bench.CLS_BIAS, tame=True)
eng = E.Engine("resnet50", 1, 9, dtype=dtype); eng.load_state(state)
x = bench.synth_images(torch, B, 1000, "cuda")
if dtype == "f32": x = x.float()
for _ in range(2): eng.detect(x)
reps = 5
per = eng.profile_ops(x, reps=reps)
plan = eng._plan(B, *bench.CANVAS)
tot = 0.0
print("%-28s %9s %9s %8s  shape" % ("op", "ms", "GFLOP", "TFLOP/s"))
for (kind, ms), op in zip(per, eng.active_ops(plan) + [("detect",)]):
    ms /= reps; tot += ms
    if kind in ("conv", "dual"):
        d = op[1]; fl = bench.conv_flops(d, B)
        g = d.g[0]
        print("%-28s %9.4f %9.2f %8.1f  M=%d N=%d K=%d k%dx%d s%d groups=%d" % (op[2], ms, fl / 1e9, fl / ms / 1e9,
              sum(d.g[i].Hout * d.g[i].Wout for i in range(d.ngroups)) * B, d.N, d.KH * d.KW * d.Crun, d.KH, d.KW, d.sy, d.ngroups))
    else:
        print("%-28s %9.4f" % (kind, ms))
print("total %.3f ms" % tot)
