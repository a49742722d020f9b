"""GPU parity of the backward primitives against torch-CPU float64 autograd (parity unpinned: TF's autodiff cannot run
here; the mathematical gradient of the restated forward graph is the oracle).
Tolerances: fp32 path 1e-4 of the gradient scale (exact-f32 MFMA chains + f32 atomics over thousands of pixels);
bf16 path: inputs rounded to bf16 on both sides, 2e-2 of the scale (bf16 output rounding / f32 accumulation)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
DT = {"f32": (torch.float32, 1), "bf16": (torch.bfloat16, 0)}


def q(x, dtype):
    return x.to(DT[dtype][0]).to(torch.float64)


def tol(dtype):
    return 1e-4 if dtype == "f32" else 2e-2


def fwd_ref(x_nhwc, w_hwio, stride, pad_t, pad_l, Ho, Wo):
    kh, kw = w_hwio.shape[0], w_hwio.shape[1]
    H, W = x_nhwc.shape[1], x_nhwc.shape[2]
    pb = max((Ho - 1) * stride + kh - pad_t - H, 0)
    pr = max((Wo - 1) * stride + kw - pad_l - W, 0)
    xp = F.pad(x_nhwc.permute(0, 3, 1, 2), (pad_l, pr, pad_t, pb))
    return F.conv2d(xp, w_hwio.permute(3, 2, 0, 1), None, stride=stride)[:, :, :Ho, :Wo].permute(0, 2, 3, 1)


def pack_fwd(w_hwio, dtype):
    kh, kw, cin, cout = w_hwio.shape
    rows = -(-cout // 128) * 128
    wk = torch.zeros(rows, kh * kw * cin, dtype=torch.float64)
    wk[:cout] = w_hwio.permute(3, 0, 1, 2).reshape(cout, -1)
    return wk.to(DT[dtype][0]).to(DEV).contiguous(), rows


CASES = [  # (H, W, cin, cout, k, stride, pad)  pad: int or "same"
    (12, 17, 64, 64, 3, 1, 1),
    (9, 13, 256, 64, 1, 1, 0),
    (10, 11, 128, 256, 3, 1, "same"),
    (17, 23, 256, 128, 1, 2, 0),          # stride-2 1x1 'valid': dgrad scatters with out_step 2
    (25, 42, 512, 256, 3, 2, "same"),     # P6: dgrad on the zero-inserted dY
    (13, 21, 256, 256, 3, 2, "same"),     # P7
]


def geometry(H, W, k, stride, pad):
    if pad == "same":
        Ho, Wo = -(-H // stride), -(-W // stride)
        pt = max((Ho - 1) * stride + k - H, 0) // 2
        pl = max((Wo - 1) * stride + k - W, 0) // 2
    else:
        pt = pl = pad
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    return Ho, Wo, pt, pl


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_dgrad_and_wgrad(pkg, handle, dtype, case):
    L = pkg._lib
    H, W, cin, cout, k, stride, pad = CASES[case]
    B = 2
    tdt, code = DT[dtype]
    g = torch.Generator().manual_seed(100 + case)
    x = q(torch.randn(B, H, W, cin, generator=g, dtype=torch.float64), dtype).requires_grad_(True)
    w = q(torch.randn(k, k, cin, cout, generator=g, dtype=torch.float64) / math.sqrt(k * k * cin), dtype).requires_grad_(True)
    Ho, Wo, pt, pl = geometry(H, W, k, stride, pad)
    y = fwd_ref(x, w, stride, pt, pl, Ho, Wo)
    dy = q(torch.randn(B, Ho, Wo, cout, generator=g, dtype=torch.float64), dtype)
    other = q(torch.randn(B, H, W, cin, generator=g, dtype=torch.float64), dtype)       # gradient already accumulated
    act = torch.randn(B, H, W, cin, generator=g, dtype=torch.float64)                  # forward activation (ReLU mask source)
    dx_ref, dw_ref = torch.autograd.grad(y, [x, w], dy)
    dx_ref = (dx_ref + other) * (q(act, dtype) > 0)

    xd = x.detach().to(tdt).to(DEV).contiguous()
    dyd = dy.to(tdt).to(DEV).contiguous()
    wk, rows = pack_fwd(w.detach(), dtype)
    # ---------------- dgrad
    rows_d = -(-cin // 128) * 128
    wd = torch.empty(rows_d, k * k * cout, dtype=tdt, device=DEV)
    handle.check(L.lib.rtn_pack_dgrad_weights(handle.raw, wk.data_ptr(), wd.data_ptr(), code, cout, rows, k, k, cin, cout, rows_d))
    dxd = other.to(tdt).to(DEV).contiguous()          # in/out: accumulate
    actd = act.to(tdt).to(DEV).contiguous()
    d = L.ConvDesc()
    d.ngroups, d.batch, d.dtype = 1, B, code
    d.w, d.w_rows, d.N, d.KH, d.KW = wd.data_ptr(), rows_d, cin, k, k
    d.Crun = d.pix_stride = cout
    d.sy = d.sx = 1
    d.out_ld = cin
    d.flags = L.CONV_RES_SAME | L.CONV_RELU_MASK
    grp = L.ConvGroup()
    keep = []
    if stride == 1:
        src, Hs, Ws = dyd, Ho, Wo
        d.pad_t, d.pad_l = k - 1 - pt, k - 1 - pl
        grp.Hout, grp.Wout = H, W
    elif k == 1:
        src, Hs, Ws = dyd, Ho, Wo
        d.pad_t = d.pad_l = 0
        grp.Hout, grp.Wout = Ho, Wo
        grp.out_step, grp.out_pix_w = 2, W
        # the scattered gradient only exists on even pixels: the reference "other" contribution elsewhere stays as is,
        # so compare on the sampled pixels and check the rest is untouched
    else:
        Hu, Wu = 2 * Ho - 1, 2 * Wo - 1
        up = torch.empty(B, Hu, Wu, cout, dtype=tdt, device=DEV)
        handle.check(L.lib.rtn_zero_insert2(handle.raw, dyd.data_ptr(), up.data_ptr(), code, B, Ho, Wo, cout, Hu, Wu))
        keep.append(up)
        src, Hs, Ws = up, Hu, Wu
        d.pad_t, d.pad_l = k - 1 - pt, k - 1 - pl
        grp.Hout, grp.Wout = H, W
    grp.in_, grp.in_elems = src.data_ptr(), src.numel()
    grp.in_img_stride, grp.in_row_stride = Hs * Ws * cout, Ws * cout
    grp.Hin, grp.Win = Hs, Ws
    grp.out, grp.out_elems, grp.out_img_stride = dxd.data_ptr(), dxd.numel(), H * W * cin
    grp.res, grp.res_elems, grp.res_img_stride, grp.res_ld = dxd.data_ptr(), dxd.numel(), H * W * cin, cin
    grp.mask, grp.mask_elems, grp.mask_img_stride, grp.mask_ld = actd.data_ptr(), actd.numel(), H * W * cin, cin
    d.g[0] = grp
    handle.check(L.lib.rtn_conv2d_dgrad(handle.raw, C.byref(d)))
    torch.cuda.synchronize()
    got = dxd.cpu().double()
    scale = max(1.0, float(dx_ref.abs().max()))
    if stride == 2 and k == 1:
        err = float((got[:, ::2, ::2] - dx_ref[:, ::2, ::2]).abs().max())
        rest = torch.ones(H, W, dtype=torch.bool)
        rest[::2, ::2] = False
        assert torch.equal(got[:, rest], q(other, dtype)[:, rest])        # untouched off the stride grid
    else:
        err = float((got - dx_ref).abs().max())
    assert err <= tol(dtype) * scale, "dgrad err %.3e scale %.2f" % (err, scale)

    # ---------------- wgrad (accumulates into a pre-filled buffer) and bias grad
    d2 = L.ConvDesc()
    d2.ngroups, d2.batch, d2.dtype = 1, B, code
    d2.w_rows, d2.N, d2.KH, d2.KW = rows, cout, k, k
    d2.Crun = d2.pix_stride = cin
    d2.sy = d2.sx = stride
    d2.pad_t, d2.pad_l = pt, pl
    d2.out_ld = cout
    g2 = L.ConvGroup()
    g2.in_, g2.in_elems = xd.data_ptr(), xd.numel()
    g2.in_img_stride, g2.in_row_stride = H * W * cin, W * cin
    g2.Hin, g2.Win, g2.Hout, g2.Wout = H, W, Ho, Wo
    g2.out, g2.out_elems, g2.out_img_stride = dyd.data_ptr(), dyd.numel(), Ho * Wo * cout
    d2.g[0] = g2
    wsb = L.lib.rtn_conv2d_wgrad_workspace_bytes(C.byref(d2))
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    dW = torch.full((rows, k * k * cin), 0.5, dtype=torch.float32, device=DEV)
    dbf = torch.full((rows,), 0.25, dtype=torch.float32, device=DEV)
    handle.check(L.lib.rtn_conv2d_wgrad_bias(handle.raw, C.byref(d2), dW.data_ptr(), dbf.data_ptr(), cout, ws.data_ptr(), wsb))
    db = torch.full((rows,), 0.25, dtype=torch.float32, device=DEV)
    handle.check(L.lib.rtn_bias_grad(handle.raw, dyd.data_ptr(), code, B * Ho * Wo, cout, cout, db.data_ptr()))
    torch.cuda.synchronize()
    want = dw_ref.permute(3, 0, 1, 2).reshape(cout, -1)
    gotw = dW.cpu().double()
    scale = max(1.0, float(want.abs().max()))
    err = float((gotw[:cout] - 0.5 - want).abs().max())
    assert err <= tol(dtype) * scale, "wgrad err %.3e scale %.2f" % (err, scale)
    assert torch.all(gotw[cout:] == 0.5)
    wantb = dy.sum(dim=(0, 1, 2))
    errb = float((db.cpu().double()[:cout] - 0.25 - wantb).abs().max())
    assert errb <= tol(dtype) * max(1.0, float(wantb.abs().max()))
    errf = float((dbf.cpu().double()[:cout] - 0.25 - wantb).abs().max())          # BiasAddGrad fused into wgrad
    assert errf <= tol(dtype) * max(1.0, float(wantb.abs().max())) and torch.all(dbf.cpu()[cout:] == 0.25)


@pytest.mark.parametrize("mode", ["res", "mask_pre", "res+mask", "res+mask_pre", "mask"])
@pytest.mark.parametrize("impl,levels,cin,cout,k,B,grid", [
    (4, [(16, 24), (8, 12), (4, 6), (2, 3), (1, 2)], 256, 256, 3, 2, 0),     # head-tower data gradient: five levels, one grouped launch
    (4, [(40, 67)], 136, 256, 3, 3, 3),       # N = 136 gradient channels (columns 136..255 of the tile never stored), 3 workgroups
    (4, [(40, 67), (20, 34)], 128, 128, 3, 2, 2),     # res3 branch2b data gradient: the 128-column instance (8-byte residual / mask loads)
    (5, [(25, 42)], 256, 128, 1, 8, 0),       # 1x1: gradient of a branch2c layer (K = 128 bytes x 2, N = 256)
    (5, [(33, 50)], 512, 256, 1, 2, 2),       # two N tiles, 2 workgroups walk all tiles
])
def test_dgrad_on_the_persistent_kernels(pkg, handle, monkeypatch, impl, levels, cin, cout, k, B, grid, mode):
    """The data-gradient launches of the training step on generations 4 / 5 (csrc/rtn_conv_halo8.hip, rtn_conv_gemm8.hip): their
    register epilogue reads the accumulated gradient (RTN_CONV_RES_SAME, in place) and the forward activation (RTN_CONV_RELU_MASK,
    before or after the add) 16 bytes per lane.  dX = mask * (dY (*) W^T) + other  |  mask * (dY (*) W^T + other), against float64
    autograd on the bf16-rounded operands; the launch must really have been the generation asked for."""
    L = pkg._lib
    dtype = "bf16"
    tdt, code = DT[dtype]
    monkeypatch.setenv("RTN_CONV_IMPL", str(impl))
    monkeypatch.setenv("RTN_CONV_H8_GRID", str(grid))
    g = torch.Generator().manual_seed(300 + impl + grid)
    w = q(torch.randn(k, k, cin, cout, generator=g, dtype=torch.float64) / math.sqrt(k * k * cin), dtype)
    wk, rows = pack_fwd(w, dtype)
    rows_d = (256 if cin > 128 else 128) if impl == 4 else cin
    wd = torch.zeros(rows_d, k * k * cout, dtype=tdt, device=DEV)
    handle.check(L.lib.rtn_pack_dgrad_weights(handle.raw, wk.data_ptr(), wd.data_ptr(), code, cout, rows, k, k, cin, cout, rows_d))
    d = L.ConvDesc()
    d.ngroups, d.batch, d.dtype = len(levels), B, code
    d.w, d.w_rows, d.N, d.KH, d.KW = wd.data_ptr(), rows_d, cin, k, k
    d.Crun = d.pix_stride = cout
    d.sy = d.sx = 1
    d.out_ld = cin
    d.pad_t = d.pad_l = (k - 1) // 2
    d.flags = (L.CONV_RES_SAME if "res" in mode else 0) | (L.CONV_RELU_MASK if "mask" in mode else 0) | \
        (L.CONV_MASK_PRE if "pre" in mode else 0)
    keep, outs, wants = [], [], []
    for gi, (H, W) in enumerate(levels):
        x = q(torch.randn(B, H, W, cin, generator=g, dtype=torch.float64), dtype).requires_grad_(True)
        y = fwd_ref(x, w, 1, (k - 1) // 2, (k - 1) // 2, H, W)
        dy = q(torch.randn(B, H, W, cout, generator=g, dtype=torch.float64), dtype)
        other = q(torch.randn(B, H, W, cin, generator=g, dtype=torch.float64), dtype)
        act = q(torch.randn(B, H, W, cin, generator=g, dtype=torch.float64), dtype)
        act[0, 0, 0, :8] = torch.tensor([0.0, -0.0, 1e-30, -1e-30, 1.0, -1.0, 0.5, 0.0])     # zeros of either sign do not pass
        dx = torch.autograd.grad(y, x, dy)[0]
        keepm = (act > 0) if "mask" in mode else torch.ones_like(act, dtype=torch.bool)
        if "res" in mode:
            dx = dx * keepm + other if "pre" in mode else (dx + other) * keepm
        else:
            dx = dx * keepm
        dyd = dy.to(tdt).to(DEV).contiguous()
        dxd = other.to(tdt).to(DEV).contiguous() if "res" in mode else torch.full((B, H, W, cin), -77.0, dtype=tdt, device=DEV)
        actd = act.to(tdt).to(DEV).contiguous()
        grp = L.ConvGroup()
        grp.in_, grp.in_elems = dyd.data_ptr(), dyd.numel()
        grp.in_img_stride, grp.in_row_stride = H * W * cout, W * cout
        grp.Hin, grp.Win, grp.Hout, grp.Wout = H, W, H, W
        grp.out, grp.out_elems, grp.out_img_stride = dxd.data_ptr(), dxd.numel(), H * W * cin
        if "res" in mode:
            grp.res, grp.res_elems, grp.res_img_stride, grp.res_ld = dxd.data_ptr(), dxd.numel(), H * W * cin, cin
        if "mask" in mode:
            grp.mask, grp.mask_elems, grp.mask_img_stride, grp.mask_ld = actd.data_ptr(), actd.numel(), H * W * cin, cin
        d.g[gi] = grp
        keep += [dyd, actd]
        outs.append(dxd)
        wants.append(dx)
    handle.check(L.lib.rtn_conv2d_dgrad(handle.raw, C.byref(d)))
    torch.cuda.synchronize()
    assert L.lib.rtn_debug_last_conv_impl(handle.raw) == impl
    for o, want in zip(outs, wants):
        err = float((o.cpu().double() - want).abs().max())
        assert err <= tol(dtype) * max(1.0, float(want.abs().max())), "dgrad err %.3e" % err



@pytest.mark.parametrize("levels,cin,cout,B,bias", [
    ([(16, 24), (8, 12), (4, 6), (2, 3), (1, 2)], 256, 256, 4, True),     # head tower: five levels, runs change level inside a split
    ([(100, 167)], 256, 256, 1, True),         # P3 at 800 x 1333: the widest window (D = 6), 32 splits
    ([(25, 42)], 256, 256, 2, False),          # P5 / res4 branch2b-like, no bias
    ([(40, 67)], 128, 128, 3, True),           # res3 branch2b: 2 output tiles, 128 splits asked for
    ([(13, 21)], 512, 512, 3, False),          # res5 branch2b: 32 output tiles, 8 splits
    ([(7, 250)], 64, 128, 1, True),            # image rows of 250 pixels: D = 4, the widest the LDS holds; one channel tile
    ([(3, 5), (64, 64)], 128, 256, 2, True),   # a tiny level first: its stages end inside the first split
    ([(40, 334)], 64, 64, 2, False),           # res2 branch2b: the 64-filter form, D = 6 (160 KiB of LDS), one output tile
    ([(16, 24), (8, 12), (4, 6), (2, 3), (1, 2)], 256, 64, 4, "concat"),     # head output: dY padded to 64 columns, levels inside one tensor
    ([(30, 40), (15, 20)], 128, 256, 2, "concat"),                           # the same layout on the 128-filter form
    ([(1, 7)], 64, 128, 40, True),             # one-row images: every slot's upper and lower neighbours are padding
    ([(5, 1)], 128, 64, 48, True),             # one-column images (padded rows of 2 slots), 64-filter form
])
def test_wgrad_window_kernel(pkg, handle, monkeypatch, levels, cin, cout, B, bias):
    """csrc/rtn_wgrad_win.hip: weight (+ bias) gradient of the stride-1 3x3 layers with all nine taps in one output tile over a
    sliding window of the input.  Against float64 autograd on the bf16-rounded operands, against the general kernels
    (RTN_WGRAD_WIN=0), and bit-for-bit against itself on a second launch (ordered slab sums).
    bias == "concat": the levels' dY lie one after another inside one [B, cells of all levels, cout] tensor (the head outputs)."""
    L = pkg._lib
    dtype = "bf16"
    tdt, code = DT[dtype]
    concat = bias == "concat"
    total = sum(H * W for H, W in levels)
    dyfull = torch.zeros(B, total, cout, dtype=tdt, device=DEV) if concat else None
    off = 0
    g = torch.Generator().manual_seed(900 + cin + cout)
    w = q(torch.randn(3, 3, cin, cout, generator=g, dtype=torch.float64) / math.sqrt(9 * cin), dtype).requires_grad_(True)
    d = L.ConvDesc()
    d.ngroups, d.batch, d.dtype = len(levels), B, code
    d.w_rows, d.N, d.KH, d.KW = cout, cout, 3, 3
    d.Crun = d.pix_stride = cin
    d.sy = d.sx = 1
    d.pad_t = d.pad_l = 1
    d.out_ld = cout
    keep, want, wantb = [], 0, 0
    for gi, (H, W) in enumerate(levels):
        x = q(torch.randn(B, H, W, cin, generator=g, dtype=torch.float64), dtype)
        dy = q(torch.randn(B, H, W, cout, generator=g, dtype=torch.float64), dtype)
        y = fwd_ref(x, w, 1, 1, 1, H, W)
        want = want + torch.autograd.grad(y, w, dy)[0]
        wantb = wantb + dy.sum(dim=(0, 1, 2))
        xd, dyd = x.to(tdt).to(DEV).contiguous(), dy.to(tdt).to(DEV).contiguous()
        keep += [xd, dyd]
        grp = L.ConvGroup()
        grp.in_, grp.in_elems = xd.data_ptr(), xd.numel()
        grp.in_img_stride, grp.in_row_stride = H * W * cin, W * cin
        grp.Hin, grp.Win, grp.Hout, grp.Wout = H, W, H, W
        if concat:
            dyfull[:, off:off + H * W, :] = dyd.reshape(B, H * W, cout)
            grp.out, grp.out_elems, grp.out_img_stride, grp.out_off = dyfull.data_ptr(), dyfull.numel(), total * cout, off * cout
            off += H * W
        else:
            grp.out, grp.out_elems, grp.out_img_stride = dyd.data_ptr(), dyd.numel(), H * W * cout
        d.g[gi] = grp
    wantm = want.permute(3, 0, 1, 2).reshape(cout, -1)

    def run(win):
        monkeypatch.setenv("RTN_WGRAD_WIN", "1" if win else "0")
        wsb = L.lib.rtn_conv2d_wgrad_workspace_bytes(C.byref(d))
        ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
        dW = torch.full((cout, 9 * cin), 0.5, dtype=torch.float32, device=DEV)
        db = torch.full((cout,), 0.25, dtype=torch.float32, device=DEV)
        if bias:
            handle.check(L.lib.rtn_conv2d_wgrad_bias(handle.raw, C.byref(d), dW.data_ptr(), db.data_ptr(), cout, ws.data_ptr(), wsb))
        else:
            handle.check(L.lib.rtn_conv2d_wgrad(handle.raw, C.byref(d), dW.data_ptr(), ws.data_ptr(), wsb))
        torch.cuda.synchronize()
        assert (L.lib.rtn_debug_last_wgrad_impl(handle.raw) == 4) == win       # 0 / 2 / 3: the general kernels
        return dW.cpu(), db.cpu()

    dW1, db1 = run(True)
    scale = max(1.0, float(wantm.abs().max()))
    err = float((dW1.double() - 0.5 - wantm).abs().max())
    assert err <= tol(dtype) * scale, "wgrad err %.3e scale %.2f" % (err, scale)
    if bias:
        errb = float((db1.double() - 0.25 - wantb).abs().max())
        assert errb <= tol(dtype) * max(1.0, float(wantb.abs().max())), "bias grad err %.3e" % errb
    else:
        assert float((db1 - 0.25).abs().max()) == 0.0
    dW2, db2 = run(True)
    assert torch.equal(dW1, dW2) and torch.equal(db1, db2)                 # ordered sums: the same bits every time
    dW0, db0 = run(False)                                                   # the general kernel: same values, other order
    assert float((dW0 - dW1).abs().max()) <= 1e-3 * scale
    assert float((db0 - db1).abs().max()) <= 1e-3 * max(1.0, float(wantb.abs().max())) if bias else True


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("H,W,cin,cout,k,B,env", [
    (40, 67, 256, 64, 1, 4, {}),                           # 128 x 128 kernel, many pixel splits (XCD map)
    (40, 67, 64, 64, 3, 4, {}),                            # 3x3, taps
    (50, 84, 256, 256, 3, 2, {"RTN_WGRAD_DMA": "2"}),      # the 256 x 256 LDS-DMA kernel
    (25, 42, 512, 2048, 1, 2, {}),                         # many output tiles, few splits
])
def test_wgrad_general_kernels_repeat_bit_for_bit(pkg, handle, monkeypatch, dtype, H, W, cin, cout, k, B, env):
    """The pixel splits of the general weight-gradient kernels go to slabs behind the row-info table and are added in split order:
    no float atomics, two runs give the same bits (and the same values as the float-atomic path, RTN_WGRAD_SLAB=0)."""
    L = pkg._lib
    tdt, code = DT[dtype]
    for kk, v in env.items():
        monkeypatch.setenv(kk, v)
    monkeypatch.setenv("RTN_WGRAD_WIN", "0")               # the general kernels are the subject here
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, H, W, cin, generator=g).to(tdt).to(DEV).contiguous()
    dy = torch.randn(B, H, W, cout, generator=g).to(tdt).to(DEV).contiguous()
    d = L.ConvDesc()
    d.ngroups, d.batch, d.dtype = 1, B, code
    d.w_rows, d.N, d.KH, d.KW = cout, cout, k, k
    d.Crun = d.pix_stride = cin
    d.sy = d.sx = 1
    d.pad_t = d.pad_l = k // 2
    d.out_ld = cout
    grp = L.ConvGroup()
    grp.in_, grp.in_elems, grp.in_img_stride, grp.in_row_stride = x.data_ptr(), x.numel(), H * W * cin, W * cin
    grp.Hin, grp.Win, grp.Hout, grp.Wout = H, W, H, W
    grp.out, grp.out_elems, grp.out_img_stride = dy.data_ptr(), dy.numel(), H * W * cout
    d.g[0] = grp

    def run(slab):
        monkeypatch.setenv("RTN_WGRAD_SLAB", "1" if slab else "0")
        wsb = L.lib.rtn_conv2d_wgrad_workspace_bytes(C.byref(d))
        ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
        dW = torch.full((cout, k * k * cin), 0.5, dtype=torch.float32, device=DEV)
        db = torch.full((cout,), 0.25, dtype=torch.float32, device=DEV)
        handle.check(L.lib.rtn_conv2d_wgrad_bias(handle.raw, C.byref(d), dW.data_ptr(), db.data_ptr(), cout, ws.data_ptr(), wsb))
        torch.cuda.synchronize()
        return dW.cpu(), db.cpu()

    w1, b1 = run(True)
    w2, b2 = run(True)
    assert torch.equal(w1, w2) and torch.equal(b1, b2)
    w0, b0 = run(False)
    scale = max(1.0, float(w0.abs().max()))
    assert float((w0 - w1).abs().max()) <= 1e-4 * scale and float((b0 - b1).abs().max()) <= 1e-4 * max(1.0, float(b0.abs().max()))


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_wgrad_grouped_levels_and_padded_head_output(pkg, handle, dtype):
    """Shared head weights: one wgrad launch over five pyramid levels; dY of the 36-channel output padded to 64."""
    L = pkg._lib
    tdt, code = DT[dtype]
    B, cin, cout, cp = 2, 256, 36, 64
    levels = [(16, 24), (8, 12), (4, 6), (2, 3), (1, 2)]
    g = torch.Generator().manual_seed(7)
    w = q(torch.randn(3, 3, cin, cout, generator=g, dtype=torch.float64) / 48, dtype).requires_grad_(True)
    total_cells = sum(h * wd for h, wd in levels)
    dy32 = torch.randn(B, total_cells, cout, generator=g)                       # f32 loss gradient, concatenated levels
    dyp = torch.empty(B, total_cells, cp, dtype=tdt, device=DEV)
    dy32d = dy32.to(DEV)
    handle.check(L.lib.rtn_pad_cast_rows(handle.raw, dy32d.data_ptr(), dyp.data_ptr(), code, B * total_cells, cout, cp))
    d = L.ConvDesc()
    d.ngroups, d.batch, d.dtype = len(levels), B, code
    d.w_rows, d.N, d.KH, d.KW = 128, cp, 3, 3
    d.Crun = d.pix_stride = cin
    d.sy = d.sx = 1
    d.pad_t = d.pad_l = 1
    d.out_ld = cp
    keep, want, off = [], 0, 0
    for gi, (H, W) in enumerate(levels):
        x = q(torch.randn(B, H, W, cin, generator=g, dtype=torch.float64), dtype)
        y = fwd_ref(x, w, 1, 1, 1, H, W)
        dyl = q(dy32[:, off:off + H * W].double(), dtype).reshape(B, H, W, cout)
        want = want + torch.autograd.grad(y, w, dyl)[0]
        xd = x.to(tdt).to(DEV).contiguous()
        keep.append(xd)
        grp = L.ConvGroup()
        grp.in_, grp.in_elems = xd.data_ptr(), xd.numel()
        grp.in_img_stride, grp.in_row_stride = H * W * cin, W * cin
        grp.Hin, grp.Win, grp.Hout, grp.Wout = H, W, H, W
        grp.out, grp.out_elems = dyp.data_ptr(), dyp.numel()
        grp.out_img_stride, grp.out_off = total_cells * cp, off * cp
        d.g[gi] = grp
        off += H * W
    wsb = L.lib.rtn_conv2d_wgrad_workspace_bytes(C.byref(d))
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    dW = torch.zeros(128, 9 * cin, dtype=torch.float32, device=DEV)
    handle.check(L.lib.rtn_conv2d_wgrad(handle.raw, C.byref(d), dW.data_ptr(), ws.data_ptr(), wsb))
    torch.cuda.synchronize()
    wantm = want.permute(3, 0, 1, 2).reshape(cout, -1)
    got = dW.cpu().double()
    assert float((got[:cout] - wantm).abs().max()) <= tol(dtype) * max(1.0, float(wantm.abs().max()))
    assert torch.all(got[cout:] == 0)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [((25, 42), (50, 84)), ((50, 84), (100, 167)), ((33, 25), (65, 50))])
def test_upsample_add_bwd(pkg, handle, dtype, shape):
    L = pkg._lib
    tdt, code = DT[dtype]
    (Hs, Ws), (Hd, Wd) = shape
    B, Cc = 2, 64
    g = torch.Generator().manual_seed(3)
    dd = q(torch.randn(B, Hd, Wd, Cc, generator=g, dtype=torch.float64), dtype)
    prev = q(torch.randn(B, Hs, Ws, Cc, generator=g, dtype=torch.float64), dtype)
    ys = np.minimum(np.floor(np.arange(Hd, dtype=np.float32) * (np.float32(Hs) / np.float32(Hd))).astype(np.int64), Hs - 1)
    xs = np.minimum(np.floor(np.arange(Wd, dtype=np.float32) * (np.float32(Ws) / np.float32(Wd))).astype(np.int64), Ws - 1)
    want = torch.zeros(B, Hs, Ws, Cc, dtype=torch.float64)
    want.index_put_((torch.arange(B)[:, None, None], torch.as_tensor(ys)[None, :, None], torch.as_tensor(xs)[None, None, :]), dd, accumulate=True)
    for acc in (0, 1):
        out = prev.to(tdt).to(DEV).contiguous()
        ddd = dd.to(tdt).to(DEV).contiguous()
        handle.check(L.lib.rtn_upsample_add_bwd(handle.raw, ddd.data_ptr(), out.data_ptr(), code, B, Hd, Wd, Hs, Ws, Cc, acc))
        torch.cuda.synchronize()
        w = want + (prev if acc else 0)
        assert float((out.cpu().double() - w).abs().max()) <= (1e-5 if dtype == "f32" else 4e-2) * max(1.0, float(w.abs().max()))


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_maxpool_bwd(pkg, handle, dtype):
    L = pkg._lib
    tdt, code = DT[dtype]
    B, H, W, Cc = 2, 21, 34, 64
    g = torch.Generator().manual_seed(4)
    x = torch.relu(q(torch.randn(B, H, W, Cc, generator=g, dtype=torch.float64), dtype))      # ReLU output: many exact zeros/ties
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    dy = q(torch.randn(B, Ho, Wo, Cc, generator=g, dtype=torch.float64), dtype)
    pth, ptw = max((Ho - 1) * 2 + 3 - H, 0), max((Wo - 1) * 2 + 3 - W, 0)
    xr = x.clone().requires_grad_(True)
    xp = F.pad(xr.permute(0, 3, 1, 2), (ptw // 2, ptw - ptw // 2, pth // 2, pth - pth // 2), value=float("-inf"))
    y = F.max_pool2d(xp, 3, 2).permute(0, 2, 3, 1)
    want = torch.autograd.grad(y, xr, dy)[0] * (x > 0)
    xd, dyd = x.to(tdt).to(DEV).contiguous(), dy.to(tdt).to(DEV).contiguous()
    dx = torch.empty_like(xd)
    scratch = torch.empty(B * H * W * Cc, dtype=torch.float32, device=DEV)
    handle.check(L.lib.rtn_maxpool3x3s2_tfsame_bwd(handle.raw, xd.data_ptr(), dyd.data_ptr(), dx.data_ptr(), code, B, H, W, Cc, scratch.data_ptr(), 1))
    torch.cuda.synchronize()
    assert float((dx.cpu().double() - want).abs().max()) <= (1e-5 if dtype == "f32" else 4e-2) * max(1.0, float(want.abs().max()))
    # training-mode pair: forward records the winning tap, backward is an atomic-free gather — same answers
    yd = torch.empty(B, Ho, Wo, Cc, dtype=tdt, device=DEV)
    idx = torch.empty(B * Ho * Wo * Cc, dtype=torch.uint8, device=DEV)
    handle.check(L.lib.rtn_maxpool3x3s2_tfsame_fwd_idx(handle.raw, xd.data_ptr(), yd.data_ptr(), idx.data_ptr(), code, B, H, W, Cc))
    dx2 = torch.full_like(xd, 3.0)
    handle.check(L.lib.rtn_maxpool3x3s2_tfsame_bwd_idx(handle.raw, dyd.data_ptr(), idx.data_ptr(), xd.data_ptr(), dx2.data_ptr(), code, B, H, W, Cc, 1))
    torch.cuda.synchronize()
    assert torch.equal(yd.cpu().double(), y.detach())
    assert float((dx2.cpu().double() - want).abs().max()) <= (1e-5 if dtype == "f32" else 4e-2) * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("fwd", ["f32", "bf16"])
def test_adam_clipnorm_matches_keras_formula(pkg, handle, fwd):
    L = pkg._lib
    n = 100003
    g = torch.Generator().manual_seed(5)
    w0 = torch.randn(n, generator=g)
    grad = torch.randn(n, generator=g) * 0.01
    gscale = torch.rand(n, generator=g) + 0.5
    gscale[::17] = 0.0                                   # frozen / structurally-zero slots
    fold = torch.rand(n, generator=g) + 0.5
    lr, b1, b2, eps, clipnorm, gm = 1e-4, 0.9, 0.999, 1e-7, 0.001, 0.5
    wd, md, vd = w0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    gd, gsd, fd = grad.to(DEV), gscale.to(DEV), fold.to(DEV)
    tdt, code = DT[fwd]
    wf = torch.empty(n, dtype=tdt, device=DEV)
    ss = torch.zeros(1, dtype=torch.float64, device=DEV)
    wsb = L.lib.rtn_sumsq_workspace_bytes()
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    # float64 restatement of keras.optimizers.Adam.get_updates + clip_norm (global norm)
    w, m, v = w0.double(), torch.zeros(n, dtype=torch.float64), torch.zeros(n, dtype=torch.float64)
    for step in (1, 2, 3):
        handle.check(L.lib.rtn_sumsq(handle.raw, gd.data_ptr(), gsd.data_ptr(), n, ss.data_ptr(), ws.data_ptr(), wsb))
        handle.check(L.lib.rtn_adam_clipnorm_step(handle.raw, wd.data_ptr(), md.data_ptr(), vd.data_ptr(), gd.data_ptr(), gsd.data_ptr(),
                                                  fd.data_ptr(), wf.data_ptr(), code, n, step, lr, b1, b2, eps, ss.data_ptr(), clipnorm, gm))
        ge = grad.double() * gscale.double() * gm
        norm = float(torch.sqrt((ge ** 2).sum()))
        assert abs(float(ss.item()) - float(((grad.double() * gscale.double()) ** 2).sum())) <= 1e-6 * float(ss.item())
        if norm > clipnorm:
            ge = ge * (clipnorm / norm)
        lr_t = lr * math.sqrt(1 - b2 ** step) / (1 - b1 ** step)
        m = b1 * m + (1 - b1) * ge
        v = b2 * v + (1 - b2) * ge ** 2
        w = w - lr_t * m / (torch.sqrt(v) + eps)
    torch.cuda.synchronize()
    assert float((wd.cpu().double() - w).abs().max()) <= 2e-6
    assert torch.equal(wd.cpu()[::17], w0[::17])                                   # zero-scaled slots never move
    wantf = (wd.cpu() * fold).to(tdt)
    assert float((wf.cpu().float() - wantf.float()).abs().max()) <= (1e-6 if fwd == "f32" else 1e-2)


@pytest.mark.parametrize("case", [2, 4, 5])
def test_wgrad_dma_kernel_cases(pkg, handle, monkeypatch, case):
    """RTN_WGRAD_DMA=2 sends every bf16 layer with more than 128 filters to the 256 x 256 LDS-DMA wgrad kernel (by default only
    layers with >= 2048 pixel tiles take it): the same cases as above, incl. stride 2 and the fused bias gradient."""
    monkeypatch.setenv("RTN_WGRAD_DMA", "2")
    monkeypatch.setenv("RTN_WGRAD_WIN", "0")
    test_dgrad_and_wgrad(pkg, handle, "bf16", case)


def test_wgrad_dma_grouped_levels_match_the_small_kernel(pkg, handle, monkeypatch):
    """A head hidden layer (256 -> 256, 3x3, five pyramid levels in one launch, bias gradient fused): the two wgrad kernels
    must agree to f32 summation order (both add bf16 x bf16 products in f32; stated 1e-4 of the largest gradient)."""
    L = pkg._lib
    tdt, code = DT["bf16"]
    B, cin, cout = 3, 256, 256
    levels = [(23, 31), (12, 16), (6, 8), (3, 4), (2, 2)]
    g = torch.Generator().manual_seed(17)
    total = sum(h * w for h, w in levels)
    dy = torch.randn(B, total, cout, generator=g).to(tdt).to(DEV).contiguous()
    d = L.ConvDesc()
    d.ngroups, d.batch, d.dtype = len(levels), B, code
    d.w_rows, d.N, d.KH, d.KW = 256, cout, 3, 3
    d.Crun = d.pix_stride = cin
    d.sy = d.sx = 1
    d.pad_t = d.pad_l = 1
    d.out_ld = cout
    keep, off = [], 0
    for gi, (H, W) in enumerate(levels):
        xd = torch.randn(B, H, W, cin, generator=g).to(tdt).to(DEV).contiguous()
        keep.append(xd)
        grp = L.ConvGroup()
        grp.in_, grp.in_elems = xd.data_ptr(), xd.numel()
        grp.in_img_stride, grp.in_row_stride = H * W * cin, W * cin
        grp.Hin, grp.Win, grp.Hout, grp.Wout = H, W, H, W
        grp.out, grp.out_elems = dy.data_ptr(), dy.numel()
        grp.out_img_stride, grp.out_off = total * cout, off * cout
        d.g[gi] = grp
        off += H * W
    wsb = L.lib.rtn_conv2d_wgrad_workspace_bytes(C.byref(d))
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    got = {}
    monkeypatch.setenv("RTN_WGRAD_WIN", "0")
    for mode in ("0", "2"):
        monkeypatch.setenv("RTN_WGRAD_DMA", mode)
        dW = torch.zeros(256, 9 * cin, dtype=torch.float32, device=DEV)
        db = torch.zeros(256, dtype=torch.float32, device=DEV)
        handle.check(L.lib.rtn_conv2d_wgrad_bias(handle.raw, C.byref(d), dW.data_ptr(), db.data_ptr(), cout, ws.data_ptr(), wsb))
        torch.cuda.synchronize()
        got[mode] = (dW.cpu().double(), db.cpu().double())
    for a, b in zip(got["0"], got["2"]):
        scale = float(a.abs().max())
        assert scale > 1.0 and float((a - b).abs().max()) <= 1e-4 * scale
    want_db = dy.float().cpu().double().sum(dim=(0, 1))
    assert float((got["2"][1] - want_db).abs().max()) <= 1e-4 * float(want_db.abs().max())


def test_pack_dgrad_multi_equals_per_layer_pack(pkg, handle):
    """rtn_pack_dgrad_weights_multi (every layer in one launch, table-driven) must write the bits of rtn_pack_dgrad_weights."""
    L = pkg._lib
    tdt, code = DT["bf16"]
    g = torch.Generator().manual_seed(3)
    layers = [(64, 3, 3, 64, 64), (256, 1, 1, 64, 256), (36, 3, 3, 256, 64), (128, 1, 1, 256, 128)]     # cout, kh, kw, cin, dY channels
    rows, total, keep, want = [], 0, [], []
    for cout, kh, kw, cin, crun in layers:
        rows_f = -(-cout // 128) * 128
        wf = torch.randn(rows_f, kh * kw * cin, generator=g).to(tdt).to(DEV).contiguous()
        rows_d = -(-cin // 128) * 128
        wd = torch.full((rows_d, kh * kw * crun), 7.0, dtype=tdt, device=DEV)
        ref = torch.empty_like(wd)
        handle.check(L.lib.rtn_pack_dgrad_weights(handle.raw, wf.data_ptr(), ref.data_ptr(), code, cout, rows_f, kh, kw, cin, crun, rows_d))
        rows.append([wf.data_ptr(), wd.data_ptr(), cout, kh, kw, cin, crun, rows_d, total, 0])
        total += wd.numel()
        keep += [wf, wd]
        want.append((wd, ref))
    table = torch.tensor(rows, dtype=torch.int64, device=DEV)
    handle.check(L.lib.rtn_pack_dgrad_weights_multi(handle.raw, table.data_ptr(), len(rows), total, code))
    torch.cuda.synchronize()
    for wd, ref in want:
        assert torch.equal(wd, ref)



@pytest.mark.parametrize("H,W,cin,cout,B,grid,mode", [
    (33, 51, 256, 512, 3, 0, "res+mask"),      # res3a_branch1 / res4a_branch2a-like: N = 256 gradient channels, K = 512
    (17, 23, 256, 128, 2, 2, "res+mask_pre"),  # two workgroups, odd extents: the last row / column of the stride grid
    (40, 66, 128, 256, 2, 3, "res+mask"),      # N = 128: the 128-column instance (8-byte residual / mask accesses) with the scatter
    (21, 35, 512, 1024, 1, 0, "mask"),         # no accumulated gradient: plain scatter of masked values
])
def test_stride2_1x1_dgrad_scatter_on_generation5(pkg, handle, monkeypatch, H, W, cin, cout, B, grid, mode):
    """The data gradient of a stride-2 1x1 'valid' convolution (the first block of a stage: res3a/4a/5a branch1 and branch2a,
    keras_resnet bottleneck behind model/defineModel.py:376-380) is a GEMM over dY whose rows land on every second pixel of the
    input grid (rtn_conv_group_t.out_step = 2).  Generation 5's register epilogue now does that scatter itself - output, accumulated
    gradient (in place) and ReLU-mask rows are all addressed through the (b, 2 oy, 2 ox) pixel of the row - instead of leaving
    these layers to generation 2.  Against float64 autograd on the bf16 operands; pixels off the stride grid keep their bits."""
    L = pkg._lib
    dtype = "bf16"
    tdt, code = DT[dtype]
    monkeypatch.setenv("RTN_CONV_IMPL", "5")
    monkeypatch.setenv("RTN_CONV_H8_GRID", str(grid))
    g = torch.Generator().manual_seed(700 + grid + cin)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    x = q(torch.randn(B, H, W, cin, generator=g, dtype=torch.float64), dtype).requires_grad_(True)
    w = q(torch.randn(1, 1, cin, cout, generator=g, dtype=torch.float64) / math.sqrt(cin), dtype)
    y = fwd_ref(x, w, 2, 0, 0, Ho, Wo)
    dy = q(torch.randn(B, Ho, Wo, cout, generator=g, dtype=torch.float64), dtype)
    other = q(torch.randn(B, H, W, cin, generator=g, dtype=torch.float64), dtype)
    act = q(torch.randn(B, H, W, cin, generator=g, dtype=torch.float64), dtype)
    dx = torch.autograd.grad(y, x, dy)[0]
    keepm = act > 0
    if "res" in mode:
        want = dx * keepm + other if "pre" in mode else (dx + other) * keepm
    else:
        want = dx * keepm
    wk, rows = pack_fwd(w, dtype)
    wd = torch.zeros(cin, cout, dtype=tdt, device=DEV)
    handle.check(L.lib.rtn_pack_dgrad_weights(handle.raw, wk.data_ptr(), wd.data_ptr(), code, cout, rows, 1, 1, cin, cout, cin))
    dyd, actd = dy.to(tdt).to(DEV).contiguous(), act.to(tdt).to(DEV).contiguous()
    start = other if "res" in mode else torch.full((B, H, W, cin), -77.0, dtype=torch.float64)
    dxd = start.to(tdt).to(DEV).contiguous()
    d = L.ConvDesc()
    d.ngroups, d.batch, d.dtype = 1, B, code
    d.w, d.w_rows, d.N, d.KH, d.KW = wd.data_ptr(), cin, cin, 1, 1
    d.Crun = d.pix_stride = cout
    d.sy = d.sx = 1
    d.out_ld = cin
    d.flags = (L.CONV_RES_SAME if "res" in mode else 0) | L.CONV_RELU_MASK | (L.CONV_MASK_PRE if "pre" in mode else 0)
    grp = L.ConvGroup()
    grp.in_, grp.in_elems, grp.in_img_stride, grp.in_row_stride = dyd.data_ptr(), dyd.numel(), Ho * Wo * cout, Wo * cout
    grp.Hin, grp.Win, grp.Hout, grp.Wout = Ho, Wo, Ho, Wo
    grp.out_step, grp.out_pix_w = 2, W
    grp.out, grp.out_elems, grp.out_img_stride = dxd.data_ptr(), dxd.numel(), H * W * cin
    if "res" in mode:
        grp.res, grp.res_elems, grp.res_img_stride, grp.res_ld = dxd.data_ptr(), dxd.numel(), H * W * cin, cin
    grp.mask, grp.mask_elems, grp.mask_img_stride, grp.mask_ld = actd.data_ptr(), actd.numel(), H * W * cin, cin
    d.g[0] = grp
    handle.check(L.lib.rtn_conv2d_dgrad(handle.raw, C.byref(d)))
    torch.cuda.synchronize()
    assert L.lib.rtn_debug_last_conv_impl(handle.raw) == 5
    got = dxd.cpu().double()
    scale = max(1.0, float(want.abs().max()))
    err = float((got[:, ::2, ::2] - want[:, ::2, ::2]).abs().max())
    assert err <= tol(dtype) * scale, "scattered dgrad err %.3e scale %.2f" % (err, scale)
    rest = torch.ones(H, W, dtype=torch.bool)
    rest[::2, ::2] = False
    assert torch.equal(got[:, rest], q(start, dtype)[:, rest])            # untouched off the stride grid
