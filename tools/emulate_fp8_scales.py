"""CPU emulation of the fp8 plan (engine.calibrate_fp8(backbone=True)) on the random ResNet-101 network: is its error dominated by the
3-bit mantissa of e4m3 or by the choice of scales?  Variants: per-tensor weight scale (the engine's), per-output-channel weight
scale, per-level input scale of the towers, and 'ideal scales' (per-channel weights + per-level + per-layer activation scales)."""
import sys, importlib
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle.ref_net import RefNet
import oracle.ref_net as RN
Wt = importlib.import_module("retinanet-for-table-detection_amd.weights")

def q8(x, scale):
    return (torch.clamp(x * scale, -448, 448).to(torch.float8_e4m3fn).to(x.dtype)) / scale

class Fp8Net(RefNet):
    def __init__(self, *a, per_channel=False, per_level=False, margin=1.25, **k):
        super().__init__(*a, **k)
        self.pc, self.pl, self.margin = per_channel, per_level, margin
        self.calib, self.amax = True, {}
    def is8(self, name):
        if name.startswith(("pyramid_regression_", "pyramid_classification_")) or name == "P3":
            return True
        if name.endswith("_branch2b"):
            return self.s[name + "/kernel"].shape[2] >= 128
        return False
    def conv(self, x, name, **kw):
        if not self.is8(name):
            return super().conv(x, name, **kw)
        key = name if not (name.startswith("pyramid_") and not self.pl) else name          # per tensor per layer; towers share over levels unless per_level
        lvl = x.shape[2]
        k2 = (name, lvl) if self.pl else name
        if self.calib:
            self.amax[k2] = max(self.amax.get(k2, 0.0), float(x.abs().max()))
            self.amax[name] = max(self.amax.get(name, 0.0), float(x.abs().max()))
            return super().conv(x, name, **kw)
        sx = 448.0 / (self.margin * max(self.amax[k2 if self.pl else name], 1e-20))
        xq = q8(x, sx)
        w = self.s[name + "/kernel"]
        bn = kw.get("bn")
        wf = w
        if bn is not None:
            g, v = self.s[bn + "/gamma"], self.s[bn + "/moving_variance"]
            wf = w * (g / torch.sqrt(v + RN.BN_EPS)).view(1, 1, 1, -1)
        if self.pc:
            sw = 448.0 / wf.abs().amax(dim=(0, 1, 2)).clamp_min(1e-30)
            wq = q8(wf, sw.view(1, 1, 1, -1))
        else:
            wq = q8(wf, 448.0 / float(wf.abs().max()))
        # run the base conv with the quantised tensors: put the folded, quantised kernel in place of the original and drop the BN scale
        saved = dict(self.s)
        self.s[name + "/kernel"] = wq
        if bn is not None:
            g, beta, m, v = [self.s[bn + s_] for s_ in ("/gamma", "/beta", "/moving_mean", "/moving_variance")]
            scale = g / torch.sqrt(v + RN.BN_EPS)
            self.s[bn + "/gamma"] = torch.sqrt(v + RN.BN_EPS)            # scale becomes 1
            self.s[bn + "/beta"] = beta - m * scale + m                   # shift stays beta - m*scale
        y = super().conv(xq, name, **kw)
        self.s = saved
        return y

canvas = (384, 384)
state = Wt.init_state("resnet101", 1, 9, seed=4, randomize_bn=True, cls_bias=0.0, tame=True)
g = torch.Generator().manual_seed(5)
raw = torch.clamp(torch.empty(1, canvas[0], canvas[1], 3).exponential_(1 / 12.0, generator=g) * torch.rand(1, canvas[0], canvas[1], 3, generator=g), 0, 255).round()
x = (raw / 127.5 - 1.0).numpy()
ref_r, ref_c = RefNet(state, backbone="resnet101", dtype=torch.float64).forward(x)
emu_r, emu_c = RefNet(state, backbone="resnet101", dtype=torch.float32, emulate_bf16=True).forward(x)
rel = lambda a: float(torch.sqrt(((a.double() - ref_r) ** 2).mean()) / torch.sqrt((ref_r ** 2).mean()))
print("bf16 emulation: regression rel-RMS %.4f" % rel(emu_r), flush=True)
for pc, pl in ((False, False), (True, False), (False, True), (True, True)):
    net = Fp8Net(state, backbone="resnet101", dtype=torch.float32, per_channel=pc, per_level=pl)
    net.forward(x)
    net.calib = False
    r, c = net.forward(x)
    print("fp8 plan, weight scales per %s, tower input scales per %s: regression rel-RMS %.4f, score max drift %.4f" % (
        "channel" if pc else "tensor", "level" if pl else "tensor", rel(r), float((c.double() - ref_c).abs().max())), flush=True)
