"""Parity of the device network against the oracle at the sizes BASELINE.json benches, with every default tile / lane / fusion
choice (no knob set): configs[1] (ResNet-50, 800x1333, batch 8, bf16 — and the fp32 parity path on the same canvas) and
configs[4] (ResNet-101, 1024x1024: bf16 and the fp8 plan).  The small-canvas tests of test_gpu_net.py reach the full-size tile
selections (256x256 tiles over ~700 row tiles, multi-round grids, the tail split at res4, split-K at P6/P7 sizes) only through
forced knobs; here they are the ones the launcher picks by itself.

Oracle: oracle/ref_net.py (torch-CPU float64, and float32 with bf16 emulation).  Parity unpinned (no Keras/TF fixture exists in
the reference), see tests/test_gpu_net.py.  The device runs the whole batch; the oracle, which needs seconds per image, checks the
FIRST and the LAST image of it (the last image's rows sit in the last, partly filled tiles of every layer).

Stated tolerances, boxes in pixels of the canvas after RegressBoxes (x = anchor + 0.2 * side * delta, model/layers.py:107-150):
  fp32 path   every decoded box within 1e-3 px (BASELINE.json north_star) on every level, P6/P7 included (measured <= 5.0e-4 px);
              scores within 1e-5.
  bf16 path   scores within 2e-2 of both oracles (the bf16 noise floor, test_gpu_net.py); |box drift| <= 1.5 % of the anchor side
              on every level (a regression-delta error of 7.5e-2: the drift is 0.2 x side x delta error, so it GROWS with the
              level's anchors) and, in pixels, <= 2 px on P3..P5 and <= 4 px on P6/P7, whose anchors are up to 575 / 1149 px long.
              Measured on MI355X (image 0 vs float64): round-1 kernels P3 0.62, P4 0.99, P5 1.41, P6 2.37, P7 2.18 px = 0.96 /
              0.69 / 0.57 / 0.46 / 0.25 % of the side, scores 9.6e-3; with the persistent kernels and the fused bottleneck blocks
              (another f32 summation order, same rounding points) 0.60 / 0.89 / 1.31 / 2.49 / 1.93 px = 1.01 / 0.69 / 0.60 / 0.46 /
              0.25 %, scores 1.0e-2: the same noise, another realisation of it; with K-sliced stage-5 layers 0.56 / 0.89 / 1.49 /
              2.32 / 2.51 px.  Those bounds are against the FLOAT64 oracle (the truth).  The bf16-emulating torch oracle is a second
              bf16 computation whose own distance from the truth is of the same size (printed as "yardstick"), so two bf16 results can
              be 1.5 x further apart than either is from the truth: against it the bounds are 1.5 x the ones above (measured 0.62 /
              1.28 / 2.19 / 2.28 / 3.31 px), and the engine may not be further from the truth than 1.5 x the yardstick + 0.25 px.
  ResNet-101  (configs[4], 1024x1024) bf16: absolute bounds 4.5 / 7.5 / 11.5 / 16 / 25 px on P3..P7, 7 % of the anchor side, scores
              6.5e-2, regression rms 0.06 (that random 101-layer network amplifies rounding 4 x more than the ResNet-50 one: torch-CPU's
              own bf16 emulation of the graph, the yardstick, loses 2.8 px / 4 % of the side on P3), and within 1.5 x the yardstick;
              fp8 plan (towers + backbone 3x3 + P3 in e4m3) against the FLOAT64 oracle: regression relative RMS <= 0.10, score
              rms <= 0.06 / max <= 0.45, box drift <= 0.5 of the anchor side (measured 0.080, 0.35 max, 0.42): what 39 e4m3 layers in
              a row cost on random filters, stated as such.
Per-level drifts are printed (run with -s)."""
import importlib

import numpy as np
import pytest
import torch

from oracle import ref_numpy as R
from oracle.ref_net import RefNet

pytestmark = pytest.mark.gpu


def mods(pkg):
    return importlib.import_module(pkg.__name__ + ".engine"), importlib.import_module(pkg.__name__ + ".weights")


def pages(B, canvas, seed):
    g = torch.Generator().manual_seed(seed)
    raw = torch.clamp(torch.empty(B, canvas[0], canvas[1], 3).exponential_(1 / 12.0, generator=g) *
                      torch.rand(B, canvas[0], canvas[1], 3, generator=g), 0, 255).round()
    return raw.to(torch.uint8)


def level_slices(canvas):
    out, o = [], 0
    for (h, w) in R.level_shapes(canvas + (3,)):
        out.append(slice(o, o + h * w * 9))
        o += h * w * 9
    return out


def drift_report(tag, reg, cls, oreg, ocls, canvas):
    """max |box drift| in px and as a fraction of the anchor side, per pyramid level; max score drift."""
    a32 = R.anchors_f32(canvas + (3,))
    side = np.stack([a32[:, 2] - a32[:, 0], a32[:, 3] - a32[:, 1], a32[:, 2] - a32[:, 0], a32[:, 3] - a32[:, 1]], axis=1).astype(np.float64)
    got = R.decode_boxes_f32(a32, reg, canvas).astype(np.float64)
    want = R.decode_boxes_f32(a32, oreg.astype(np.float32), canvas).astype(np.float64)
    d = np.abs(got - want)
    rows = []
    for lv, sl in zip((3, 4, 5, 6, 7), level_slices(canvas)):
        rows.append((lv, float(d[sl].max()), float((d[sl] / side[sl]).max()), float(side[sl].max())))
    dcls = float(np.abs(cls - ocls).max())
    print("%s: " % tag + "  ".join("P%d %.3e px (%.2e of side, side<=%.0f)" % r for r in rows) + "  | score %.3e" % dcls)
    return rows, dcls


@pytest.fixture(scope="module")
def r50_case(pkg):
    """State, a batch of 8 pages at 800x1333 and the float64 / bf16-emulating oracle outputs of its first and last image."""
    E, Wt = mods(pkg)
    canvas = (800, 1333)
    state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=0.0, tame=True)
    u8 = pages(8, canvas, seed=77)
    x = torch.as_tensor(R.preprocess_custom_tf(u8.numpy()))
    picks = (0, 7)
    o64, oemu = {}, {}
    for b in picks:
        xb = x[b:b + 1].numpy()
        r, c = RefNet(state, dtype=torch.float64).forward(xb)
        o64[b] = (r.numpy()[0], c.numpy()[0])
        r, c = RefNet(state, dtype=torch.float32, emulate_bf16=True).forward(xb)
        oemu[b] = (r.numpy()[0], c.numpy()[0])
    return {"state": state, "x": x, "u8": u8, "canvas": canvas, "picks": picks, "o64": o64, "oemu": oemu}


def test_r50_800x1333_batch8_bf16_against_both_oracles(pkg, r50_case):
    E, _ = mods(pkg)
    c = r50_case
    eng = E.Engine("resnet50", 1, 9, dtype="bf16")
    eng.load_state(c["state"])
    assert eng.two_streams and eng.fuse_stem and eng.fuse_shortcut          # the benched configuration
    reg, cls = eng.forward(c["x"].cuda())
    torch.cuda.synchronize()
    reg, cls = reg.cpu().numpy(), cls.cpu().numpy()
    assert reg.shape == (8, 200700, 4) and cls.shape == (8, 200700, 1)
    for b in c["picks"]:
        # the float64 oracle is the truth: <= 2 px on P3-P5, <= 4 px on P6 / P7, <= 1.5 % of the anchor's side, scores <= 2e-2.
        # The bf16-emulating oracle (torch CPU, every conv input / weight / output rounded to bf16) is a second bf16 computation with
        # its own rounding noise of the same size (printed below as "yardstick"): against it the bound is 1.5 x the one above, and
        # the engine may not be further from the truth than 1.5 x the yardstick is (+ 0.25 px).
        yard, _ = drift_report("yardstick: bf16-emulating oracle of image %d vs float64 oracle" % b, c["oemu"][b][0], c["oemu"][b][1],
                               c["o64"][b][0], c["o64"][b][1], c["canvas"])
        for tag, (oreg, ocls), slack in (("float64 oracle", c["o64"][b], 1.0), ("bf16-emulating oracle", c["oemu"][b], 1.5)):
            rows, dcls = drift_report("bf16 image %d vs %s" % (b, tag), reg[b], cls[b], oreg, ocls, c["canvas"])
            assert dcls <= 2e-2 * slack
            for (lv, dpx, dfrac, _), (_, ypx, _, _) in zip(rows, yard):
                assert dpx <= slack * (2.0 if lv <= 5 else 4.0) and dfrac <= slack * 1.5e-2, "P%d drifts %.3f px (%.3e of the anchor side)" % (lv, dpx, dfrac)
                if slack == 1.0:
                    assert dpx <= 1.5 * ypx + 0.25, "P%d: engine %.3f px from the truth, torch bf16 emulation %.3f px" % (lv, dpx, ypx)
    # the uint8 entry of the same pages (normalisation fused into the packer) is what a data loader hands over: same bits
    reg8, cls8 = eng.forward(c["u8"].cuda())
    torch.cuda.synchronize()
    assert np.array_equal(reg8.cpu().numpy(), reg) and np.array_equal(cls8.cpu().numpy(), cls)
    # post-processing at this size: bit-exact on the engine's own head outputs
    boxes, scores, labels = eng.detect(c["x"].cuda())
    torch.cuda.synchronize()
    a32 = R.anchors_f32(c["canvas"] + (3,))
    for b in c["picks"]:
        wb, ws, wl = R.filter_detections(R.decode_boxes_f32(a32, reg[b], c["canvas"]), cls[b])
        assert np.array_equal(boxes[b].cpu().numpy(), wb) and np.array_equal(scores[b].cpu().numpy(), ws)
        assert np.array_equal(labels[b].cpu().numpy(), wl)


def test_r50_800x1333_fp32_within_1e3_px(pkg, r50_case):
    E, _ = mods(pkg)
    c = r50_case
    eng = E.Engine("resnet50", 1, 9, dtype="f32")
    eng.load_state(c["state"])
    x2 = c["x"][[0, 7]].contiguous()                   # the fp32 parity path at batch 2: first and last image of the batch
    reg, cls = eng.forward(x2.cuda())
    torch.cuda.synchronize()
    reg, cls = reg.cpu().numpy(), cls.cpu().numpy()
    for i, b in enumerate(c["picks"]):
        oreg, ocls = c["o64"][b]
        rows, dcls = drift_report("fp32 image %d vs float64 oracle" % b, reg[i], cls[i], oreg, ocls, c["canvas"])
        assert dcls <= 1e-5
        for lv, dpx, dfrac, smax in rows:
            assert dpx <= 1e-3, "P%d: %.3e px > 1e-3 (BASELINE.json north_star)" % (lv, dpx)          # measured <= 5.0e-4 on every level


@pytest.fixture(scope="module")
def r101_case(pkg):
    E, Wt = mods(pkg)
    canvas = (1024, 1024)
    state = Wt.init_state("resnet101", 1, 9, seed=4, randomize_bn=True, cls_bias=0.0, tame=True)
    u8 = pages(2, canvas, seed=5)
    x = torch.as_tensor(R.preprocess_custom_tf(u8.numpy()))
    r, c = RefNet(state, backbone="resnet101", dtype=torch.float64).forward(x[:1].numpy())
    er, ec = RefNet(state, backbone="resnet101", dtype=torch.float32, emulate_bf16=True).forward(x[:1].numpy())
    return {"state": state, "x": x, "canvas": canvas, "o64": (r.numpy()[0], c.numpy()[0]), "oemu": (er.numpy()[0], ec.numpy()[0])}


def test_r101_1024_bf16_and_fp8_against_the_float64_oracle(pkg, r101_case):
    """BASELINE.json configs[4]: ResNet-101-FPN on 1024x1024 pages (model/defineModel.py:376-380 resnet101 branch: 23 blocks in
    stage 4, named res4b1..res4b22), first in bf16, then with calibrate_fp8(backbone=True): towers, every 3x3 branch2b with >= 128
    channels and P3 in e4m3.  Both against the float64 oracle of image 0 (calibration sees image 1 too)."""
    E, _ = mods(pkg)
    c = r101_case
    eng = E.Engine("resnet101", 1, 9, dtype="bf16")
    eng.load_state(c["state"])
    names = [op[2] for op in eng._plan(2, 1024, 1024)["ops"] if op[0] == "conv"]
    assert "res4b22_branch2c" in names and "res3b3_branch2b" in names and "res5c_branch2a" in names
    xd = c["x"].cuda()
    reg, cls = eng.forward(xd)
    torch.cuda.synchronize()
    assert reg.shape == (2, 196416, 4)
    oreg, ocls = c["o64"]
    # This 101-layer random network amplifies bf16 rounding four times more than the ResNet-50 one (23 blocks in stage 4): the
    # yardstick is what torch-CPU's own bf16 emulation of the same graph loses against float64 (measured: 2.8 px = 4.1 % of the
    # side on P3, scores 4.2e-2, regression rms 0.045; the device, any kernel generation: 2.7 px, 4.2e-2, 0.044).  Stated bound:
    # the device stays within 1.5 x that yardstick on every level.
    yrows, ycls = drift_report("R101 yardstick: torch-CPU bf16 emulation vs float64", c["oemu"][0], c["oemu"][1], oreg, ocls, c["canvas"])
    rows, dcls = drift_report("R101 bf16 vs float64 oracle", reg[0].cpu().numpy(), cls[0].cpu().numpy(), oreg, ocls, c["canvas"])
    # ABSOLUTE bounds (what the yardstick measured on MI355X's host, x 1.6, written out): box drift <= 4.5 / 7.5 / 11.5 / 16 / 25 px
    # on P3..P7 and <= 7 % of the anchor side on every level, scores <= 6.5e-2, regression rms <= 0.06.  Measured: yardstick 2.83 /
    # 4.64 / 7.08 / 10.5 / 15.7 px (4.6 % of the side at most), scores 4.2e-2, rms 0.045; device 2.5-2.7 / 4.9-5.1 / 6.7-7.8 / 9.8-10.4 /
    # 11.2-16.4 px, 4.2-4.3e-2, 0.044.  The relative check against the yardstick of THIS run stays beside them.
    ABS_PX = {3: 4.5, 4: 7.5, 5: 11.5, 6: 16.0, 7: 25.0}
    assert dcls <= 6.5e-2 and dcls <= 1.5 * ycls
    for (lv, dpx, dfrac, _), (_, ypx, yfrac, _) in zip(rows, yrows):
        assert dpx <= ABS_PX[lv] and dfrac <= 7e-2, "P%d drifts %.3f px = %.3e of the anchor side" % (lv, dpx, dfrac)
        assert dpx <= 1.5 * ypx and dfrac <= 1.5 * yfrac, "P%d drifts %.3f px (yardstick %.3f)" % (lv, dpx, ypx)
    rms_dev = float(np.sqrt(((reg[0].cpu().numpy() - oreg) ** 2).mean()))
    rms_emu = float(np.sqrt(((c["oemu"][0] - oreg) ** 2).mean()))
    print("R101 regression rms error: device %.5f, emulation %.5f" % (rms_dev, rms_emu))
    assert rms_dev <= 0.06 and rms_dev <= 1.25 * rms_emu
    rows_bf16 = rows
    # ---- fp8 plan
    eng.calibrate_fp8([xd], backbone=True)
    plan = eng._plan(2, 1024, 1024)
    n8 = sum(1 for op in eng.active_ops(plan) if op[0] == "conv8")
    assert plan["fp8"] and n8 >= 8 + 30 + 1, "fp8 ops: %d" % n8          # 8 tower layers, 30 branch2b layers, P3
    reg8, cls8 = eng.forward(xd)
    torch.cuda.synchronize()
    r8, c8 = reg8[0].cpu().numpy(), cls8[0].cpu().numpy()
    rows, dcls = drift_report("R101 fp8 vs float64 oracle", r8, c8, oreg, ocls, c["canvas"])
    rel_rms = float(np.sqrt(((r8 - oreg) ** 2).mean()) / np.sqrt((oreg ** 2).mean()))
    rms8 = float(np.sqrt(((r8 - oreg) ** 2).mean()))
    srms8 = float(np.sqrt(((c8 - ocls) ** 2).mean()))
    print("R101 fp8: regression relative RMS %.4f (rms %.5f = %.2f x the bf16 path's), scores: max drift %.4f, rms %.5f" %
          (rel_rms, rms8, rms8 / rms_dev, dcls, srms8))
    # e4m3 keeps 3 mantissa bits (6 % element error per layer, 39 such layers in a row on random filters whose sums do not average
    # it down).  Stated bounds against FLOAT64, with the measurement on MI355X beside them: regression relative RMS <= 0.10 (0.080:
    # 8.6 x the bf16 path), score rms <= 0.06 and max <= 0.45 (0.35), box drift <= 0.5 of the anchor side (0.42 on P3 = 24 px: the
    # regression values of this random network are O(5), so 8 % of them is tens of pixels).  A statement of what the fp8 plan costs
    # on this network, not a tight parity result: a trained checkpoint is what the plan has to be judged on (none exists here).
    assert rel_rms <= 0.10 and srms8 <= 0.06 and dcls <= 0.45
    for lv, dpx, dfrac, _ in rows:
        assert dfrac <= 0.5, "P%d drifts %.3e of the anchor side" % (lv, dfrac)
    boxes, scores, labels = eng.detect(xd)
    torch.cuda.synchronize()
    a32 = R.anchors_f32(c["canvas"] + (3,))
    wb, ws, wl = R.filter_detections(R.decode_boxes_f32(a32, r8, c["canvas"]), c8)
    assert np.array_equal(boxes[0].cpu().numpy(), wb) and np.array_equal(scores[0].cpu().numpy(), ws)
