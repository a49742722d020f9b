"""GPU parity for the HBM-bound kernels: stem packing, max-pool, ReLU, anchors, anchor targets, losses.
Integer/index work and everything the reference computes in NumPy is checked BIT-EXACT against the
oracle and the reference-generated golden vectors; float32 reductions have their tolerance stated."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_numpy as R
from helpers import load_case

pytestmark = pytest.mark.gpu
DEV = "cuda"


def engine_mod(pkg):
    import importlib
    return importlib.import_module(pkg.__name__ + ".engine")


# ------------------------------------------------------------------ stem pack / pool / relu
@pytest.mark.parametrize("src", ["f32", "bf16", "u8"])
@pytest.mark.parametrize("dst", ["f32", "bf16"])
def test_stem_pack(pkg, handle, src, dst):
    L = pkg._lib
    B, H, W = 2, 37, 53
    g = torch.Generator().manual_seed(1)
    if src == "u8":
        x = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
        want_in = torch.as_tensor(R.preprocess_custom_tf(x.numpy()))
        code = 2
    else:
        x = torch.randn(B, H, W, 3, generator=g)
        if src == "bf16":
            x = x.to(torch.bfloat16)
        want_in = x.float()
        code = 1 if src == "f32" else 0
    H1, W1 = (H + 1) // 2, (W + 1) // 2
    Hp, Wp = max(H + 6, 2 * (H1 - 1) + 8), max(W + 6, 2 * (W1 - 1) + 8)
    Wp += Wp & 1
    ddt = torch.float32 if dst == "f32" else torch.bfloat16
    out = torch.full((B, Hp, Wp, 4), 5.0, dtype=ddt, device=DEV)
    xd = x.to(DEV)
    handle.check(L.lib.rtn_stem_pack(handle.raw, xd.data_ptr(), code, out.data_ptr(), 1 if dst == "f32" else 0, B, H, W, Hp, Wp))
    torch.cuda.synchronize()
    want = torch.zeros(B, Hp, Wp, 4)
    want[:, 3:3 + H, 3:3 + W, :3] = want_in
    want = want.to(ddt).float()
    assert torch.equal(out.cpu().float(), want)           # bit-exact (same single rounding)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("hw", [(20, 33), (21, 34), (7, 8)])
def test_maxpool_tfsame(pkg, handle, dtype, hw):
    L = pkg._lib
    B, (H, W), Cc = 2, hw, 64
    tdt = torch.float32 if dtype == "f32" else torch.bfloat16
    x = torch.randn(B, H, W, Cc, generator=torch.Generator().manual_seed(2)).to(tdt)
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    out = torch.empty(B, Ho, Wo, Cc, dtype=tdt, device=DEV)
    xd = x.to(DEV)
    handle.check(L.lib.rtn_maxpool3x3s2_tfsame_fwd(handle.raw, xd.data_ptr(), out.data_ptr(), 1 if dtype == "f32" else 0, B, H, W, Cc))
    torch.cuda.synchronize()
    pth, ptw = max((Ho - 1) * 2 + 3 - H, 0), max((Wo - 1) * 2 + 3 - W, 0)
    xp = F.pad(x.float().permute(0, 3, 1, 2), (ptw // 2, ptw - ptw // 2, pth // 2, pth - pth // 2), value=float("-inf"))
    want = F.max_pool2d(xp, 3, 2).permute(0, 2, 3, 1)
    assert torch.equal(out.cpu().float(), want)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_relu(pkg, handle, dtype):
    L = pkg._lib
    tdt = torch.float32 if dtype == "f32" else torch.bfloat16
    x = torch.randn(3, 5, 7, 256, generator=torch.Generator().manual_seed(3)).to(tdt)
    xd = x.to(DEV)
    out = torch.empty_like(xd)
    handle.check(L.lib.rtn_relu(handle.raw, xd.data_ptr(), out.data_ptr(), 1 if dtype == "f32" else 0, x.numel()))
    torch.cuda.synchronize()
    assert torch.equal(out.cpu().float(), torch.relu(x.float()))


# ------------------------------------------------------------------ anchors
@pytest.mark.parametrize("shape", [(800, 1333), (1028, 800), (64, 96), (37, 53)])
def test_anchors_bit_exact(pkg, handle, shape):
    E = engine_mod(pkg)
    cfg, N = E.make_anchor_cfg(shape)
    want64 = R.anchors_for_shape(shape + (3,))
    assert N == want64.shape[0]
    a64 = torch.empty(N, 4, dtype=torch.float64, device=DEV)
    a32 = torch.empty(N, 4, dtype=torch.float32, device=DEV)
    handle.check(pkg.lib.rtn_anchors_f64(handle.raw, C.byref(cfg), a64.data_ptr()))
    handle.check(pkg.lib.rtn_anchors_f32(handle.raw, C.byref(cfg), a32.data_ptr()))
    torch.cuda.synchronize()
    assert np.array_equal(a64.cpu().numpy(), want64)
    assert np.array_equal(a32.cpu().numpy(), R.anchors_f32(shape + (3,)))


# ------------------------------------------------------------------ anchor targets
def run_targets(pkg, handle, canvas, shapes, gts, labels=None, K=1):
    E = engine_mod(pkg)
    cfg, N = E.make_anchor_cfg(canvas)
    B = len(shapes)
    gb = np.zeros((B, 64, 4), np.float64)
    gl = np.zeros((B, 64), np.int32)
    gc = np.zeros((B,), np.int32)
    for i, g in enumerate(gts):
        gb[i, :len(g)] = g
        gc[i] = len(g)
        if labels is not None:
            gl[i, :len(g)] = labels[i]
    hw = np.asarray(shapes, np.int32)
    t = [torch.as_tensor(a).to(DEV) for a in (gb, gl, gc, hw)]
    reg = torch.full((B, N, 5), 9.0, dtype=torch.float32, device=DEV)
    lab = torch.full((B, N, K + 1), 9.0, dtype=torch.float32, device=DEV)
    handle.check(pkg.lib.rtn_anchor_targets(handle.raw, C.byref(cfg), B, K, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(),
                                            t[3].data_ptr(), 0.4, 0.5, reg.data_ptr(), lab.data_ptr()))
    torch.cuda.synchronize()
    return reg.cpu().numpy(), lab.cpu().numpy()


@pytest.mark.parametrize("name", ["kat3", "kat2_img800x1000", "empty", "rand0", "rand1", "rand2", "small_tie"])
def test_anchor_targets_vs_reference_golden(pkg, handle, golden, name):
    canvas, shapes, gts = load_case(golden, name)
    reg, lab = run_targets(pkg, handle, canvas, shapes, gts)
    # golden index sets come from the reference's own anchor_targets_bbox
    for b in range(len(shapes)):
        pos = np.nonzero(reg[b, :, 4] == 1)[0]
        assert np.array_equal(pos, golden["tgt_%s_pos%d" % (name, b)])
        assert np.array_equal(np.nonzero(reg[b, :, 4] == -1)[0], golden["tgt_%s_ign%d" % (name, b)])
        assert np.array_equal(reg[b, pos, :4], golden["tgt_%s_regpos%d" % (name, b)])
        assert np.array_equal(lab[b, pos, 0], golden["tgt_%s_labpos%d" % (name, b)])
    # and the complete tensors are bit-identical to the (golden-pinned) oracle
    anchors = R.anchors_for_shape(canvas + (3,))
    wreg, wlab = R.anchor_targets(anchors, shapes, gts, [np.zeros(len(g)) for g in gts], 1)
    assert np.array_equal(reg, wreg) and np.array_equal(lab, wlab)


def test_anchor_targets_multiclass_batch8(pkg, handle):
    rng = np.random.RandomState(5)
    canvas, B, K = (800, 1333), 8, 3
    shapes, gts, labs = [], [], []
    for _ in range(B):
        g = rng.randint(0, 7)
        w, h = rng.uniform(80, 900, g), rng.uniform(60, 600, g)
        x1, y1 = rng.uniform(0, 1333 - w), rng.uniform(0, 800 - h)
        gts.append(np.stack([x1, y1, x1 + w, y1 + h], axis=1).reshape(-1, 4))
        labs.append(rng.randint(0, K, g))
        shapes.append((int(rng.randint(600, 801)), int(rng.randint(900, 1334))))
    reg, lab = run_targets(pkg, handle, canvas, shapes, gts, labs, K)
    wreg, wlab = R.anchor_targets(R.anchors_for_shape(canvas + (3,)), shapes, gts, labs, K)
    assert np.array_equal(reg, wreg) and np.array_equal(lab, wlab)


# ------------------------------------------------------------------ losses
def make_loss_inputs(B, N, K, seed):
    rng = np.random.RandomState(seed)
    state = rng.choice([-1.0, 0.0, 1.0], size=(B, N), p=[0.1, 0.85, 0.05]).astype(np.float32)
    lab = np.zeros((B, N, K + 1), np.float32)
    lab[..., K] = state
    cls_idx = rng.randint(0, K, size=(B, N))
    for k in range(K):
        lab[..., k] = ((state == 1) & (cls_idx == k)).astype(np.float32)
    p = rng.uniform(0.001, 0.999, size=(B, N, K)).astype(np.float32)
    p[0, :4, 0] = [0.0, 1.0, 1e-9, 1 - 1e-9]                 # saturated probabilities hit the epsilon clip
    regt = np.zeros((B, N, 5), np.float32)
    regt[..., :4] = rng.normal(size=(B, N, 4))
    regt[..., 4] = state
    pred = (regt[..., :4] + rng.normal(scale=0.15, size=(B, N, 4))).astype(np.float32)
    return lab, regt, p, pred


@pytest.mark.parametrize("B,N,K", [(2, 5000, 1), (8, 200700, 1), (2, 3001, 3)])
def test_retina_loss_fwd_bwd(pkg, handle, B, N, K):
    lab, regt, p, pred = make_loss_inputs(B, N, K, seed=N % 97)
    d = [torch.as_tensor(a).to(DEV) for a in (lab, regt, p, pred)]
    rows = B * N
    wsb = pkg.lib.rtn_retina_loss_workspace_bytes(rows)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    sums = torch.zeros(4, dtype=torch.float64, device=DEV)
    handle.check(pkg.lib.rtn_retina_loss_fwd(handle.raw, rows, K, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(),
                                             0.25, 2.0, 3.0, sums.data_ptr(), ws.data_ptr(), wsb))
    torch.cuda.synchronize()
    s1 = sums.cpu().numpy().copy()
    fc, npos, gcls = R.focal_loss(lab, p, grad=True)
    rl, nposr, greg = R.smooth_l1_loss(regt, pred, grad=True)
    # float32 per-term evaluation vs the float64 oracle: 2e-5 relative on the sums; counts exact
    assert abs(s1[0] - fc) <= 2e-5 * abs(fc) and abs(s1[1] - rl) <= 2e-5 * abs(rl)
    assert s1[2] == npos and s1[3] == nposr
    # deterministic: a second launch gives identical bits
    handle.check(pkg.lib.rtn_retina_loss_fwd(handle.raw, rows, K, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(),
                                             0.25, 2.0, 3.0, sums.data_ptr(), ws.data_ptr(), wsb))
    torch.cuda.synchronize()
    assert np.array_equal(sums.cpu().numpy(), s1)
    # backward (w.r.t. probabilities and w.r.t. logits)
    inv = 1.0 / max(1, npos)
    for wrt_logits in (0, 1):
        dcls = torch.empty(B, N, K, dtype=torch.float32, device=DEV)
        dreg = torch.empty(B, N, 4, dtype=torch.float32, device=DEV)
        handle.check(pkg.lib.rtn_retina_loss_bwd(handle.raw, rows, K, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(),
                                                 d[3].data_ptr(), 0.25, 2.0, 3.0, inv, inv, wrt_logits, dcls.data_ptr(),
                                                 dreg.data_ptr()))
        torch.cuda.synchronize()
        want = gcls * inv
        if wrt_logits:
            want = want * (p.astype(np.float64) * (1 - p.astype(np.float64)))
        got = dcls.cpu().numpy().astype(np.float64)
        # the float32 evaluation of 1/p near the clip is ill-conditioned: compare relative to magnitude
        assert np.all(np.abs(got - want) <= 1e-4 * np.maximum(np.abs(want), 1e-3))
        assert np.allclose(dreg.cpu().numpy(), greg * inv, rtol=1e-5, atol=1e-7)
