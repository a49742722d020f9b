"""rtn_chain1x1_fwd (csrc/rtn_chain.hip): the seam between two identity bottleneck blocks of the 128- / 256-channel stages as one
launch,  x_out = relu(conv1x1(h) + b2c + x_in);  a_out = relu(conv1x1(x_out) + b2a)  (keras_resnet bottleneck_2d behind
model/defineModel.py:376-380).  Checked (a) BIT FOR BIT against the two rtn_conv2d_fwd launches it replaces (same roundings, same
f32 summation order: bias-initialised accumulators, K ascending in MFMA steps of 32) and (b) against a float64 evaluation of the
chain on the bf16-rounded operands with x_out rounded to bf16 where the separate launches round it; stated tolerance of (b): 1e-2
of the output scale (bf16 storage: 2^-8 relative, a few ulps after two chained products)."""
import ctypes as C
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu


def bf(x):
    return x.to(torch.bfloat16).to(torch.float64)


def conv1x1(L, handle, x, out, w, b, n, k, flags, res=None):
    """[M][k] -> [M][n] through rtn_conv2d_fwd, as the engine would launch the layer (one image of M x 1 pixels)."""
    M = x.shape[0]
    d = L.ConvDesc()
    g = L.ConvGroup()
    g.in_, g.in_elems, g.out, g.out_elems = x.data_ptr(), x.numel(), out.data_ptr(), out.numel()
    g.in_img_stride, g.in_row_stride = M * k, k
    g.Hin, g.Win, g.Hout, g.Wout = M, 1, M, 1
    g.out_img_stride = M * n
    if res is not None:
        g.res, g.res_elems, g.res_img_stride, g.res_ld, g.Hres, g.Wres = res.data_ptr(), res.numel(), M * n, n, M, 1
    d.g[0] = g
    d.ngroups, d.batch, d.dtype = 1, 1, L.RTN_BF16
    d.w, d.bias, d.w_rows, d.N, d.KH, d.KW = w.data_ptr(), b.data_ptr(), w.shape[0], n, 1, 1
    d.Crun = d.pix_stride = k
    d.sy = d.sx = 1
    d.out_ld, d.flags = n, flags
    ws = L.attach_conv_workspace(handle, d)
    handle.check(L.lib.rtn_conv2d_fwd(handle.raw, C.byref(d)))
    return ws


@pytest.mark.parametrize("mid,M,grid", [(128, 1, 0), (128, 37, 0), (128, 2 * 25 * 42, 0), (128, 8 * 31 * 33, 3), (128, 5000, 1), (128, 2049, 2)])
def test_chain1x1_kernel(pkg, handle, monkeypatch, mid, M, grid):
    L = pkg._lib
    if grid:                                       # RTN_CHAIN_GRID: several passes per workgroup, the filter stream wraps around
        monkeypatch.setenv("RTN_CHAIN_GRID", str(grid))
    out = 4 * mid
    g = torch.Generator().manual_seed(M + mid)
    dev = torch.device("cuda")
    h = torch.relu(torch.randn(M, mid, generator=g, dtype=torch.float64))
    x = torch.relu(torch.randn(M, out, generator=g, dtype=torch.float64))
    w2c = torch.randn(out, mid, generator=g, dtype=torch.float64) / mid ** 0.5
    w2a = torch.randn(mid, out, generator=g, dtype=torch.float64) / out ** 0.5
    b2c, b2a = torch.randn(out, generator=g, dtype=torch.float64) * 0.3, torch.randn(mid, generator=g, dtype=torch.float64) * 0.3
    hq, xq, wcq, waq = bf(h), bf(x), bf(w2c), bf(w2a)
    xo = bf(torch.relu(hq @ wcq.T + b2c.float().double() + xq))
    ao = bf(torch.relu(xo @ waq.T + b2a.float().double()))
    t16 = lambda t: t.to(torch.bfloat16).to(dev).contiguous()
    hd, xd, wcd, wad = t16(h), t16(x), t16(w2c), t16(w2a)
    bcd, bad = b2c.float().to(dev), b2a.float().to(dev)
    xout = torch.full((M, out), -7.0, dtype=torch.bfloat16, device=dev)
    aout = torch.full((M, mid), -7.0, dtype=torch.bfloat16, device=dev)
    d = L.ChainDesc()
    d.h_in, d.h_in_elems, d.x_in, d.x_in_elems = hd.data_ptr(), hd.numel(), xd.data_ptr(), xd.numel()
    d.x_out, d.x_out_elems, d.a_out, d.a_out_elems = xout.data_ptr(), xout.numel(), aout.data_ptr(), aout.numel()
    d.w2c, d.b2c, d.w2a, d.b2a = wcd.data_ptr(), bcd.data_ptr(), wad.data_ptr(), bad.data_ptr()
    d.pixels, d.mid, d.out, d.next, d.dtype = M, mid, out, mid, L.RTN_BF16
    assert L.lib.rtn_chain1x1_supported(mid, out, mid) == 1 and L.lib.rtn_chain1x1_supported(64, 256, 64) == 0 and L.lib.rtn_chain1x1_supported(256, 1024, 256) == 0
    handle.check(L.lib.rtn_chain1x1_fwd(handle.raw, C.byref(d)))
    torch.cuda.synchronize()
    first = (xout.clone(), aout.clone())
    for rep in range(3):                            # the result repeats bit for bit
        handle.check(L.lib.rtn_chain1x1_fwd(handle.raw, C.byref(d)))
    torch.cuda.synchronize()
    assert torch.equal(first[0], xout) and torch.equal(first[1], aout)
    # (a) the two launches it replaces, on generation 5 (the kernel these layers take at the network's sizes)
    monkeypatch.setenv("RTN_CONV_IMPL", "5")
    x2 = torch.full((M, out), -7.0, dtype=torch.bfloat16, device=dev)
    a2 = torch.full((M, mid), -7.0, dtype=torch.bfloat16, device=dev)
    k1 = conv1x1(L, handle, hd, x2, wcd, bcd, out, mid, L.CONV_RELU | L.CONV_RES_SAME, res=xd)
    assert L.lib.rtn_debug_last_conv_impl(handle.raw) == 5
    k2 = conv1x1(L, handle, x2, a2, wad, bad, mid, out, L.CONV_RELU)
    assert L.lib.rtn_debug_last_conv_impl(handle.raw) == 5
    torch.cuda.synchronize()
    assert torch.equal(x2, xout), "x_out differs from rtn_conv2d_fwd in %d elements" % int((x2 != xout).sum())
    assert torch.equal(a2, aout), "a_out differs from rtn_conv2d_fwd in %d elements" % int((a2 != aout).sum())
    # (b) float64
    sx, sa = max(1.0, float(xo.abs().max())), max(1.0, float(ao.abs().max()))
    ex, ea = float((xout.cpu().double() - xo).abs().max()), float((aout.cpu().double() - ao).abs().max())
    print("x_out: max err %.3e of scale %.2f; a_out: %.3e of %.2f" % (ex, sx, ea, sa))
    assert ex <= 1e-2 * sx and ea <= 1.5e-2 * sa
    # error behaviour
    d.mid = 64
    assert L.lib.rtn_chain1x1_fwd(handle.raw, C.byref(d)) == -1           # not a built shape
    d.mid = mid
    d.a_out = hd.data_ptr()
    assert L.lib.rtn_chain1x1_fwd(handle.raw, C.byref(d)) == -1           # an output aliasing an input
    d.a_out = xout.data_ptr()
    assert L.lib.rtn_chain1x1_fwd(handle.raw, C.byref(d)) == -1           # two outputs in one buffer
    d.a_out = aout.data_ptr()
    d.x_in_elems = xd.numel() - 1
    assert L.lib.rtn_chain1x1_fwd(handle.raw, C.byref(d)) == -4
    d.x_in_elems = xd.numel()
    d.dtype = L.RTN_F32
    assert L.lib.rtn_chain1x1_fwd(handle.raw, C.byref(d)) == -1


def test_engine_with_fused_seams_gives_the_bits_of_separate_layers(pkg, monkeypatch):
    """Engine level (inference, bf16): with Engine.fuse_chain the op list holds one "chain" op per identity-block seam of stage 3
    (2 at ResNet-50) instead of a branch2c and a branch2a conv, and every feature map, the regression and the
    classification tensors carry the SAME BITS as with the layers launched separately on generation 5 (RTN_CONV_IMPL=5: the kernel
    those layers take at the bench size; the small canvas of this test would otherwise send them to generation 2, whose f32
    summation order differs)."""
    monkeypatch.setenv("RTN_CONV_IMPL", "5")
    E = importlib.import_module(pkg.__name__ + ".engine")
    Wt = importlib.import_module(pkg.__name__ + ".weights")
    state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=0.0, tame=True)
    g = torch.Generator().manual_seed(32)
    x = (torch.rand(2, 320, 448, 3, generator=g) * 2 - 1).cuda()
    eng = E.Engine("resnet50", 1, 9, dtype="bf16")
    eng.load_state(state)
    got = {}
    for fuse in (False, True):
        eng.fuse_chain = fuse
        plan = eng._plan(2, 320, 448)
        for t in plan["feats"]:
            t.fill_(-7.0)
        reg, cls = eng.forward(x)
        torch.cuda.synchronize()
        ops = eng.active_ops(plan)
        names = [op[2] for op in ops if op[0] in ("conv", "chain")]
        assert [op[0] for op in ops].count("chain") == (2 if fuse else 0)
        assert ("res3c_branch2a" in names) == (not fuse) and ("res3b_branch2c" in names) == (not fuse) and "res4b_branch2c" in names
        assert "res3b_branch2a" in names and "res3d_branch2c" in names       # the seam behind the projection block / the stage's end stay convs
        got[fuse] = [t.clone() for t in plan["feats"]] + [reg.clone(), cls.clone()]
    for a, b in zip(got[False], got[True]):
        assert float(a.float().abs().max()) > 0.01 and torch.equal(a, b)
    eng.training = True                              # the training forward takes the same fused seams
    assert [op[0] for op in eng.active_ops(eng._plan(2, 320, 448))].count("chain") == 2
    eng.training = False
