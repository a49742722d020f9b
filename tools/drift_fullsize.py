"""Per-level box drift of the bf16 engine against the float64 oracle at 800x1333 (batch 8, image 0) for several knob settings:
which kernel generation / fusion contributes what.   python tools/drift_fullsize.py"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_fullsize as T
from oracle.ref_net import RefNet
from oracle import ref_numpy as R
pkg = importlib.import_module("retinanet-for-table-detection_amd")
E, Wt = T.mods(pkg)
backbone = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
canvas = (800, 1333) if backbone == "resnet50" else (1024, 1024)
B = 8 if backbone == "resnet50" else 2
state = Wt.init_state(backbone, 1, 9, seed=0 if backbone == "resnet50" else 4, randomize_bn=True, cls_bias=0.0, tame=True)
u8 = T.pages(B, canvas, seed=77 if backbone == "resnet50" else 5)
x = torch.as_tensor(R.preprocess_custom_tf(u8.numpy()))
r, c = RefNet(state, backbone=backbone, dtype=torch.float64).forward(x[:1].numpy())
oreg, ocls = r.numpy()[0], c.numpy()[0]
er, ec = RefNet(state, backbone=backbone, dtype=torch.float32, emulate_bf16=True).forward(x[:1].numpy())
T.drift_report("yardstick: torch-CPU bf16 emulation vs float64", er.numpy()[0], ec.numpy()[0], oreg, ocls, canvas)
d = np.abs(er.numpy()[0] - oreg)
print("    regression delta error: max %.4f, rms %.5f" % (d.max(), np.sqrt((d ** 2).mean())))
configs = [("all new kernels", {}), ("RTN_CONV_H8=0", {"RTN_CONV_H8": "0"}), ("RTN_CONV_G8=0", {"RTN_CONV_G8": "0"}),
           ("RTN_FUSE_BOTTLENECK=0", {"RTN_FUSE_BOTTLENECK": "0"}), ("all off (round-1 kernels)", {"RTN_CONV_H8": "0", "RTN_CONV_G8": "0", "RTN_FUSE_BOTTLENECK": "0"})]
for name, env in configs:
    for k in ("RTN_CONV_H8", "RTN_CONV_G8", "RTN_FUSE_BOTTLENECK"): os.environ.pop(k, None)
    os.environ.update(env)
    eng = E.Engine(backbone, 1, 9, dtype="bf16"); eng.load_state(state)
    reg, cls = eng.forward(x.cuda()); torch.cuda.synchronize()
    T.drift_report(name, reg[0].cpu().numpy(), cls[0].cpu().numpy(), oreg, ocls, canvas)
    d = np.abs(reg[0].cpu().numpy() - oreg)
    print("    regression delta error: max %.4f, rms %.5f, 99.99th pct %.4f" % (d.max(), np.sqrt((d ** 2).mean()), np.quantile(d, 0.9999)))
