"""BASELINE.json configs[2] at its real size: one training step (forward -> focal + smooth-L1 -> backward through heads / FPN /
ResNet-50 -> global-norm clip + Adam; RetinaNet.py:125-131,280, model/losses.py:5-91, model/anchors.py:36-92) at 800x1333 with
EVERY default kernel selection (no knob set).  tests/test_gpu_train.py compares the step with autograd at 128x192, where the
full-size selections of the backward pass (256 x 256 LDS-DMA weight gradients at >= 2048 pixel tiles, the nine-tap window weight-gradient
kernel by shape, 256-split slabs, generation 4/5 data gradients with their residual / ReLU-mask epilogues) are reached only through
forced knobs; here the launcher picks them by itself and the test asserts which ones it picked.

Oracle: oracle/ref_net.train_step_oracle_per_image (torch-CPU autograd of the restated graph, one image at a time with the
merged-batch normaliser; parity unpinned as for every network-numerics test — the reference holds no fixture, TF/Keras absent).
  (a) fp32 path, batch 2 (first / last tiles of every layer) against the FLOAT64 oracle: both losses within 1e-5 relative,
      every listed layer's weight (and bias) gradient within 2e-3 of the layer's gradient scale max|g| — the bounds of the
      128x192 test, unchanged.  The two pages of the batch are the SAME page (round 4: the oracle then differentiates one image,
      32 s instead of 64; batches of different pages are (b) and tests/test_gpu_train.py).
  (b) bf16 path, batch 16 (the benched shape) against the FLOAT32 oracle (float64 costs 23 s per image on 8 cores; float32 is
      exact to 1e-6 on these sums, far below the bf16 noise being measured): losses within 3e-2 relative, per-layer gradient
      cosine >= 0.98 and norm within 10 %, two backward passes bit-identical, the kernel each layer's dgrad / wgrad ran.
      Round 4: pages 8..15 repeat pages 0..7 (one of which has no box), so the device still runs the benched batch of 16 with the
      merged-batch normaliser over all 16 pages, and the oracle - whose loss and gradients for the doubled batch are exactly those
      of the 8 distinct pages (2 x sums / 2 x positives) - differentiates 8 images instead of 16 (the GPU suite's time limit).
Measured on MI355X (round 3): (a) losses 1.2e-7 / 5.7e-7 relative, worst layer res3a_branch1 at 1.5e-4 of its gradient scale;
(b) losses 1.2e-4 / 4.8e-3 relative, worst layer pyramid_regression_0 with cosine 0.99917, norm ratios 0.990 .. 1.026.
The two tests take ~210 s, almost all of it the CPU oracle (2 float64 + 16 float32 autograd passes at 800x1333)."""
import importlib
import time

import numpy as np
import pytest
import torch

from oracle import ref_numpy as R
from oracle.ref_net import train_step_oracle_per_image

pytestmark = pytest.mark.gpu
CANVAS = (800, 1333)
LAYERS = ["conv1", "res2a_branch2a", "res2a_branch1", "res2b_branch2b", "res2c_branch2c", "res3a_branch2a", "res3a_branch1",
          "res3d_branch2b", "res4a_branch1", "res4f_branch2c", "res5a_branch2a", "res5c_branch2b", "C5_reduced", "P5", "C4_reduced",
          "P4", "C3_reduced", "P3", "P6", "P7", "pyramid_regression_0", "pyramid_regression_3", "pyramid_regression",
          "pyramid_classification_0", "pyramid_classification_2", "pyramid_classification"]


def mods(pkg):
    return [importlib.import_module(pkg.__name__ + "." + m) for m in ("engine", "weights", "trainer")]


def make_batch(B, seed):
    """Pages of the bench's pixel statistics + 1..5 table-sized ground-truth boxes per page (one page without any: its anchors are
    all background, model/anchors.py:64-80), targets from the NumPy oracle."""
    g = torch.Generator().manual_seed(seed)
    raw = torch.clamp(torch.empty(B, CANVAS[0], CANVAS[1], 3).exponential_(1 / 12.0, generator=g) *
                      torch.rand(B, CANVAS[0], CANVAS[1], 3, generator=g), 0, 255).round().to(torch.uint8)
    x = R.preprocess_custom_tf(raw.numpy())
    anchors = R.anchors_for_shape(CANVAS + (3,))
    rng = np.random.RandomState(seed)
    gts, shapes = [], []
    for b in range(B):
        n = 0 if (b == 1 and B > 2) else rng.randint(1, 6)
        w, h = rng.uniform(120, 900, n), rng.uniform(60, 500, n)
        x1, y1 = rng.uniform(0, CANVAS[1] - w), rng.uniform(0, CANVAS[0] - h)
        gts.append(np.stack([x1, y1, x1 + w, y1 + h], axis=1).reshape(n, 4))
        shapes.append((CANVAS[0], int(rng.randint(1100, CANVAS[1] + 1))))
    reg, lab = R.anchor_targets(anchors, shapes, gts, [np.zeros(len(g_)) for g_ in gts], 1)
    assert reg.shape == (B, 200700, 5) and (lab[..., 1] == 1).sum() > (50 if B > 1 else 5)
    return x, reg, lab


def unpack_grad(tr, name):
    """flat packed gradient of the folded weights -> gradient w.r.t. the Keras HWIO kernel (x fold scale)."""
    lo = tr.eng.layout[name]
    dW, db = tr.grad_views(name)
    gs = tr.gscale[lo["woff"]:lo["woff"] + lo["rows"] * lo["K"]].view(lo["rows"], lo["K"])
    g = (dW * gs).cpu().double()
    cout = lo["cout"]
    if name == "conv1":
        k = g[:cout].reshape(cout, 8, 8, 4)[:, :7, :7, :3].permute(1, 2, 3, 0)
    else:
        k = g[:cout].reshape(cout, lo["kh"], lo["kw"], lo["cin"]).permute(1, 2, 3, 0)
    return k, db[:cout].cpu().double()


def oracle(state, x, reg_t, lab_t, dtype):
    t0 = time.time()
    out = train_step_oracle_per_image(state, x, reg_t, lab_t, dtype=dtype,
                                      progress=lambda b: print("  oracle image %d done at %.0f s" % (b, time.time() - t0), flush=True))
    return out


def test_fp32_training_step_800x1333_batch2_against_float64_autograd(pkg):
    E, Wt, T = mods(pkg)
    state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=-2.0, tame=True)
    x, reg_t, lab_t = make_batch(1, seed=41)
    (l_reg, l_cls), og = oracle(state, x, reg_t, lab_t, torch.float64)
    x, reg_t, lab_t = [np.concatenate([a, a], axis=0) for a in (x, reg_t, lab_t)]      # the device runs batch 2 (see the module text)
    eng = E.Engine("resnet50", 1, 9, dtype="f32")
    eng.load_state(state)
    tr = T.Trainer(eng, lr=1e-4, clipnorm=0.001)
    sums = tr.forward_backward(torch.as_tensor(x).cuda(), torch.as_tensor(reg_t).cuda(), torch.as_tensor(lab_t).cuda())
    torch.cuda.synchronize()
    s = sums.cpu().numpy()
    got_reg, got_cls = s[1] / max(1, s[3]), s[0] / max(1, s[2])
    print("fp32 losses: regression %.8f (oracle %.8f), classification %.8f (oracle %.8f)" % (got_reg, l_reg, got_cls, l_cls))
    assert abs(got_reg - l_reg) <= 1e-5 * abs(l_reg) and abs(got_cls - l_cls) <= 1e-5 * abs(l_cls)
    worst = (0.0, "")
    for name in LAYERS:
        gk, gb = unpack_grad(tr, name)
        want = og[name + "/kernel"].double()
        scale = float(want.abs().max())
        err = float((gk - want).abs().max()) / scale
        worst = max(worst, (err, name))
        assert err <= 2e-3, "%s: weight-gradient error %.3e of its scale %.3e" % (name, err, scale)
        if name + "/bias" in og:
            wb = og[name + "/bias"].double()
            assert float((gb - wb).abs().max()) <= 2e-3 * float(wb.abs().max()), name
    print("fp32 path at 800x1333: worst layer %s, %.3e of its gradient scale" % (worst[1], worst[0]))
    tr.optimizer_step()
    torch.cuda.synchronize()
    norm = float(np.sqrt(sum(float((g_.double() ** 2).sum()) for g_ in og.values())))
    got_norm = float(torch.sqrt(tr.sumsq).item())
    assert abs(got_norm - norm) <= 2e-3 * norm, (got_norm, norm)


# Which kernel each backward op takes at batch 16 x 800 x 1333 in bf16 (rtn_debug_last_conv_impl / rtn_debug_last_wgrad_impl;
# dgrad: 2 = 256-row LDS-DMA per tap, 4 = persistent 8-phase 3x3, 5 = persistent 1x1; wgrad: 2 = 256 x 256
# LDS-DMA, 3 = 128 x 128 LDS-DMA, 4 = the nine-tap window kernel).  The cost models that choose are in csrc/rtn_conv.hip
# (conv_launch) and rtn_backward.hip (wgrad_plan, wgrad_takes_win); DESIGN.md §3.3.
EXPECTED_IMPLS = {
    # data gradients: the tower / pyramid / bottleneck 3x3 layers on the persistent 8-phase kernel, the 1x1 layers with >= 128
    # output channels on the persistent 1x1 kernel (residual + ReLU-mask epilogues), the 64-channel and stride-2 forms on generations 1-2
    ("dgrad", "pyramid_regression_0"): 4, ("dgrad", "pyramid_classification_3"): 4, ("dgrad", "P3"): 4, ("dgrad", "P4"): 4, ("dgrad", "P5"): 4,
    ("dgrad", "C3_reduced"): 5, ("dgrad", "C4_reduced"): 5, ("dgrad", "C5_reduced"): 5,
    ("dgrad", "res3b_branch2a"): 5, ("dgrad", "res3b_branch2b"): 4, ("dgrad", "res3b_branch2c"): 5,
    ("dgrad", "res4b_branch2a"): 5, ("dgrad", "res4b_branch2b"): 4, ("dgrad", "res4b_branch2c"): 5,
    ("dgrad", "res5b_branch2a"): 5, ("dgrad", "res5b_branch2b"): 2, ("dgrad", "res5b_branch2c"): 5,
    ("dgrad", "res2b_branch2b"): 2, ("dgrad", "res2b_branch2c"): 1, ("dgrad", "P6"): 2, ("dgrad", "P7"): 1,
    ("dgrad", "pyramid_regression"): 4, ("dgrad", "pyramid_classification"): 4,
    # weight gradients: the nine-tap window kernel for the stride-1 3x3 layers (towers, P3, P4, res3-res5 branch2b; its 64-filter
    # form for res2 branch2b and the head outputs), the 256 x 256 LDS-DMA kernel for the 1x1 layers with >= 2048 pixel tiles (C3_reduced, res3a_branch1), the
    # 128 x 128 LDS-DMA kernel for the rest
    ("wgrad", "pyramid_regression_0"): 4, ("wgrad", "pyramid_classification_3"): 4, ("wgrad", "P3"): 4, ("wgrad", "C3_reduced"): 2,
    ("wgrad", "res3a_branch1"): 2, ("wgrad", "P4"): 4, ("wgrad", "res3b_branch2b"): 4, ("wgrad", "res4b_branch2b"): 4,
    ("wgrad", "res4f_branch2b"): 4, ("wgrad", "res5b_branch2b"): 4, ("wgrad", "res2b_branch2b"): 4, ("wgrad", "res4b_branch2a"): 3,
    ("wgrad", "conv1"): 3, ("wgrad", "pyramid_regression"): 4, ("wgrad", "P6"): 3,
}


def test_bf16_training_step_800x1333_batch16_the_benched_shape(pkg):
    E, Wt, T = mods(pkg)
    state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=-2.0, tame=True)
    x, reg_t, lab_t = make_batch(8, seed=43)
    (l_reg, l_cls), og = oracle(state, x, reg_t, lab_t, torch.float32)
    x, reg_t, lab_t = [np.concatenate([a, a], axis=0) for a in (x, reg_t, lab_t)]      # batch 16: every page twice (see the module text)
    assert x.shape[0] == 16
    eng = E.Engine("resnet50", 1, 9, dtype="bf16")
    eng.load_state(state)
    assert eng.fuse_shortcut and eng.fuse_bottleneck
    tr = T.Trainer(eng, lr=1e-4, clipnorm=0.001)
    assert tr.wgrad_lane and tr.wgrad_lanes == 3                      # the benched configuration
    tr.record_impls = True
    xd, regd, labd = torch.as_tensor(x).cuda(), torch.as_tensor(reg_t).cuda(), torch.as_tensor(lab_t).cuda()
    sums = tr.forward_backward(xd, regd, labd)
    torch.cuda.synchronize()
    s = sums.cpu().numpy()
    got_reg, got_cls = s[1] / max(1, s[3]), s[0] / max(1, s[2])
    print("bf16 losses: regression %.6f (oracle %.6f), classification %.6f (oracle %.6f)" % (got_reg, l_reg, got_cls, l_cls))
    assert abs(got_reg - l_reg) <= 3e-2 * abs(l_reg) and abs(got_cls - l_cls) <= 3e-2 * abs(l_cls)
    g1 = tr.grad.clone()
    impls = dict(tr.impls)
    print("impls:", sorted((k[0], k[1], v) for k, v in impls.items()))
    for key, want in EXPECTED_IMPLS.items():
        assert impls.get(key) == want, "%s of %s ran kernel %s, expected %s" % (key[0], key[1], impls.get(key), want)
    worst = (0.0, "")
    for name in LAYERS:
        gk, _ = unpack_grad(tr, name)
        want = og[name + "/kernel"].double()
        cos = float((gk * want).sum() / (gk.norm() * want.norm()))
        ratio = float(gk.norm() / want.norm())
        worst = max(worst, (1 - cos, name))
        print("  %-28s cos %.5f  norm ratio %.4f" % (name, cos, ratio))
        assert cos >= 0.98 and 0.9 <= ratio <= 1.1, "%s: cos %.4f norm ratio %.3f" % (name, cos, ratio)
    print("bf16 path at 16x800x1333: worst layer %s (1 - cos = %.3e)" % (worst[1], worst[0]))
    # no float atomics anywhere in the step: a second backward pass over the same batch gives the same bits
    tr.forward_backward(xd, regd, labd)
    torch.cuda.synchronize()
    assert torch.equal(g1, tr.grad), "gradient differs between two identical passes (max %.3e)" % float((g1 - tr.grad).abs().max())
    loss = tr.train_on_batch(xd, regd, labd)
    assert np.isfinite(loss[0])
