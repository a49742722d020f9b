"""Per-op device time of one forward pass at the bench configuration (events on the launch stream).
  python tools/profile_layers.py [bf16|f32] [batch] [--fp8] [--backbone resnet101] [--canvas H W]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
PKG = bench.PKG
E = importlib.import_module(PKG + ".engine"); Wt = importlib.import_module(PKG + ".weights")
args = sys.argv[1:]
fp8 = "--fp8" in args
backbone = args[args.index("--backbone") + 1] if "--backbone" in args else "resnet50"
canvas = tuple(int(v) for v in args[args.index("--canvas") + 1:args.index("--canvas") + 3]) if "--canvas" in args else bench.CANVAS
pos = [a for i, a in enumerate(args) if not a.startswith("--") and not (i and args[i - 1] in ("--backbone", "--canvas")) and not (i > 1 and args[i - 2] == "--canvas")]
dtype = pos[0] if len(pos) > 0 else "bf16"
B = int(pos[1]) if len(pos) > 1 else bench.BATCH
state = Wt.init_state(backbone, 1, 9, seed=0, randomize_bn=True, cls_bias=bench.CLS_BIAS, tame=True)
eng = E.Engine(backbone, 1, 9, dtype=dtype); eng.load_state(state)
if canvas == bench.CANVAS:
    x = bench.synth_images(torch, B, 1000, "cuda")
else:
    x = (torch.rand(B, canvas[0], canvas[1], 3, generator=torch.Generator().manual_seed(1)) * 2 - 1).to(torch.bfloat16).cuda()
if dtype == "f32": x = x.float()
if fp8: eng.calibrate_fp8(x, backbone=True)
for _ in range(2): eng.detect(x)
reps = 5
per = eng.profile_ops(x, reps=reps)
plan = eng._plan(B, *canvas)
tot = 0.0
print("%-28s %9s %9s %8s  shape" % ("op", "ms", "GFLOP", "TFLOP/s"))
for (kind, ms), op in zip(per, eng.active_ops(plan) + [("detect",)]):
    ms /= reps; tot += ms
    if kind == "bneck":
        m = op[3]; px = m["B"] * m["H"] * m["W"]
        fl = 2.0 * px * (576 * 64 + 64 * 256 + 256 * 64 * (int(bool(m["tail"])) + int(bool(m.get("proj")))))
        by = 2.0 * px * ((64 + 64 + 256 + (64 if m["tail"] else 0)) if m.get("proj") else (64 + 256 + 256 + (64 if m["tail"] else 0)))
        print("%-28s %9.4f %9.2f %8.1f  fused bottleneck M=%d, %.0f MB -> %.2f TB/s" % (op[2], ms, fl / 1e9, fl / ms / 1e9, px, by / 1e6, by / ms / 1e9))
    elif kind == "chain":
        m = op[3]; pc = m.get("proj_c", 0); fl = 2.0 * m["pixels"] * (2 * m["mid"] + pc) * 4 * m["mid"]; by = 2.0 * m["pixels"] * ((6 * m["mid"] + pc) if pc else 10 * m["mid"])
        print("%-28s %9.4f %9.2f %8.1f  fused seam M=%d, %.0f MB -> %.2f TB/s" % (op[2], ms, fl / 1e9, fl / ms / 1e9, m["pixels"], by / 1e6, by / ms / 1e9))
    elif kind in ("conv", "dual", "conv8", "convq"):
        d = op[1]; fl = bench.conv_flops(d, B)
        g = d.g[0]
        print("%-28s %9.4f %9.2f %8.1f  M=%d N=%d K=%d k%dx%d s%d groups=%d" % (op[2] + {"conv8": " [fp8]", "convq": " [>fp8]"}.get(kind, ""), ms, fl / 1e9, fl / ms / 1e9,
              sum(d.g[i].Hout * d.g[i].Wout for i in range(d.ngroups)) * B, d.N, d.KH * d.KW * d.Crun, d.KH, d.KW, d.sy, d.ngroups))
    else:
        print("%-28s %9.4f" % (kind, ms))
print("total %.3f ms" % tot)
