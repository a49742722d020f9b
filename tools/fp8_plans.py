"""BASELINE.json configs[4] (ResNet-101-FPN, 1024 x 1024 pages): what each fp8 plan buys and what it costs.
Plans: bf16 | fp8 towers only (8 layers, 52 % of the head FLOPs, one quantisation deep) | fp8 towers + every 3x3 branch2b + P3 (39 layers).
Throughput at batch 8 (one batch at a time and two in flight), drift of image 0 of a batch of 2 against the float64 oracle."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_fullsize as T
from oracle.ref_net import RefNet
from oracle import ref_numpy as R
pkg = importlib.import_module("retinanet-for-table-detection_amd")
E, Wt = T.mods(pkg)
canvas = (1024, 1024)
state = Wt.init_state("resnet101", 1, 9, seed=4, randomize_bn=True, cls_bias=0.0, tame=True)
u8 = T.pages(2, canvas, seed=5)
x2 = torch.as_tensor(R.preprocess_custom_tf(u8.numpy()))
t0 = time.time()
r, c = RefNet(state, backbone="resnet101", dtype=torch.float64).forward(x2[:1].numpy())
oreg, ocls = r.numpy()[0], c.numpy()[0]
print("float64 oracle of image 0: %.0f s" % (time.time() - t0), flush=True)
g = torch.Generator().manual_seed(1)
x8 = (torch.rand(8, 1024, 1024, 3, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
def throughput(eng, n_in_flight):
    eng.in_flight = n_in_flight
    for _ in range(4): eng.detect(x8)
    eng.join(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): eng.detect(x8)
    eng.join(); torch.cuda.synchronize()
    eng.in_flight = 1
    return 8 * 20 / (time.perf_counter() - t)
for tag, cal in (("bf16", None), ("fp8 towers only", False), ("fp8 towers + 3x3 backbone + P3", True)):
    eng = E.Engine("resnet101", 1, 9, dtype="bf16"); eng.load_state(state)
    if cal is not None:
        eng.calibrate_fp8([x2.cuda()], backbone=cal)
    reg, cls = eng.forward(x2.cuda()); torch.cuda.synchronize()
    n8 = sum(1 for op in eng.active_ops(eng._plan(2, 1024, 1024)) if op[0] == "conv8")
    rg, cl = reg[0].cpu().numpy(), cls[0].cpu().numpy()
    rows, dcls = T.drift_report("%s vs float64" % tag, rg, cl, oreg, ocls, canvas)
    rel = float(np.sqrt(((rg - oreg) ** 2).mean()) / np.sqrt((oreg ** 2).mean()))
    if cal is not None:
        eng.calibrate_fp8([x8], backbone=cal)
    print("%-32s fp8 layers %2d | regression rel-RMS %.4f | worst box drift %.3f of the anchor side | score drift max %.3f | %.0f img/s one batch at a time, %.0f two in flight"
          % (tag, n8, rel, max(r_[2] for r_ in rows), dcls, throughput(eng, 1), throughput(eng, 2)), flush=True)
