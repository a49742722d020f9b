"""BASELINE.json configs[4] shape: ResNet-101-FPN, 1024x1024 pages, batch 8.  bf16 throughout, or with --fp8-towers the two head
towers (38 % of the FLOPs) in fp8 e4m3 after a one-batch calibration (Engine.calibrate_fp8); the backbone and FPN stay bf16.
--fp8-backbone adds every 3x3 branch2b layer with >= 128 channels (another 30 % of the FLOPs).
  python tools/bench_r101.py [resnet101|resnet152] [side] [--fp8-towers] [--fp8-backbone] [--in-flight-2]     prints images/s and the conv TFLOP/s."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, bench
E = importlib.import_module(bench.PKG + ".engine"); Wt = importlib.import_module(bench.PKG + ".weights")
fp8b = "--fp8-backbone" in sys.argv
fp8 = "--fp8-towers" in sys.argv or fp8b
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
backbone = argv[0] if len(argv) > 0 else "resnet101"
H = W = int(argv[1]) if len(argv) > 1 else 1024
B = 8
state = Wt.init_state(backbone, 1, 9, seed=0, randomize_bn=True, cls_bias=bench.CLS_BIAS, tame=True)
eng = E.Engine(backbone, 1, 9, dtype="bf16"); eng.load_state(state)
g = torch.Generator().manual_seed(1)
x = (torch.rand(B, H, W, 3, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
if fp8: eng.calibrate_fp8(x, backbone=fp8b)
if "--in-flight-2" in sys.argv: eng.in_flight = 2
for _ in range(3): eng.detect(x)
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 20
for _ in range(n): eng.detect(x)
eng.join(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
plan = eng._plan(B, H, W)
fl = sum(bench.conv_flops(op[1], B) for op in plan["ops"] if op[0] in ("conv", "conv8", "convq"))
print("%s %dx%d batch %d %s: %.3f ms/step = %.1f img/s; %.1f GFLOP/image; %.0f TFLOP/s over the step; anchors %d"
      % (backbone, H, W, B, ("bf16 + fp8 towers and 3x3 backbone layers" if fp8b else "bf16 + fp8 towers") if fp8 else "bf16", dt * 1e3, B / dt, fl / B / 1e9, fl / dt / 1e12, plan["N"]))
