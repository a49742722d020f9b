"""GPU parity: rtn_conv2d_fwd (implicit GEMM on MFMA) against a float64 torch-CPU convolution that
restates TF's semantics ('same' padding asymmetry, legacy nearest upsample, fused epilogues).
Tolerances: fp32 path 2e-5 relative to the output scale (exact-f32 MFMA, different sum order);
bf16 path compared against the same reference fed bf16-rounded inputs/weights, 1e-2 relative
(one bf16 rounding of the output, fp32 accumulation)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = {"f32": (torch.float32, 1), "bf16": (torch.bfloat16, 0)}


def q(x, dtype):
    return x.to(DT[dtype][0]).to(torch.float64)


def ref_conv(x, w_hwio, bias, stride, pad_t, pad_l, Hout, Wout):
    """x (B,H,W,C) f64, taps outside the image are zero; output extent given."""
    kh, kw = w_hwio.shape[0], w_hwio.shape[1]
    H, W = x.shape[1], x.shape[2]
    pb = max((Hout - 1) * stride + kh - pad_t - H, 0)
    pr = max((Wout - 1) * stride + kw - pad_l - W, 0)
    xp = F.pad(x.permute(0, 3, 1, 2), (pad_l, pr, pad_t, pb))
    y = F.conv2d(xp, w_hwio.permute(3, 2, 0, 1), None, stride=stride)[:, :, :Hout, :Wout]
    if bias is not None:
        y = y + bias.view(1, -1, 1, 1)
    return y.permute(0, 2, 3, 1).contiguous()


def upsample_nearest_legacy(src, oh, ow):
    ih, iw = src.shape[1], src.shape[2]
    ys = np.minimum(np.floor(np.arange(oh, dtype=np.float32) * (np.float32(ih) / np.float32(oh))).astype(np.int64), ih - 1)
    xs = np.minimum(np.floor(np.arange(ow, dtype=np.float32) * (np.float32(iw) / np.float32(ow))).astype(np.int64), iw - 1)
    return src[:, torch.as_tensor(ys)][:, :, torch.as_tensor(xs)]


def pack_w(w_hwio, dtype, dev):
    kh, kw, cin, cout = w_hwio.shape
    rows = -(-cout // 128) * 128
    wk = torch.zeros(rows, kh * kw * cin, dtype=torch.float64)
    wk[:cout] = w_hwio.permute(3, 0, 1, 2).reshape(cout, -1)
    return wk.to(DT[dtype][0]).to(dev).contiguous(), rows


LAST = {}                                          # the last run_case's descriptor / workspace (stream-K checks)


def run_case(pkg, handle, dtype, levels, cin, cout, k, stride, pad, flags=0, res_mode=None, B=2, seed=0, out_ld=None,
             concat=False, reference=True, exact=False):
    """levels: list of (H, W). Returns (got list, want list) per level (f64, NHWC).
    exact: small-integer operands - every product and every partial sum is exact in f32 whatever the order of the additions, so two
    kernels that split the K loop differently must agree bit for bit with each other and with the float64 reference."""
    L = pkg._lib
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(seed)
    tdt = DT[dtype][0]
    randn = torch.randn
    if exact:
        def randn(*shape, generator=None, dtype=None):
            return torch.randint(-3, 4, shape, generator=generator).to(dtype)
        w = torch.randint(-2, 3, (k, k, cin, cout), generator=g).double()
        bias = torch.randint(-8, 9, (cout,), generator=g).double()
    else:
        w = torch.randn(k, k, cin, cout, generator=g, dtype=torch.float64) / math.sqrt(k * k * cin)
        bias = torch.randn(cout, generator=g, dtype=torch.float64)
    wq = q(w, dtype)
    wk, rows = pack_w(w, dtype, dev)
    bk = torch.zeros(rows, dtype=torch.float32)
    bk[:cout] = bias.float()
    bk = bk.to(dev)
    d = L.ConvDesc()
    d.ngroups, d.batch, d.dtype = len(levels), B, DT[dtype][1]
    d.w, d.bias, d.w_rows, d.N, d.KH, d.KW = wk.data_ptr(), bk.data_ptr(), rows, cout, k, k
    d.Crun = d.pix_stride = cin
    d.sy = d.sx = stride
    out_f32 = bool(flags & L.CONV_OUT_F32) or dtype == "f32"
    odt = torch.float32 if out_f32 else tdt
    ld = cout if out_ld is None else out_ld
    d.out_ld = ld
    keep, wants, outs = [], [], []
    if concat:                                    # head-output style: all levels into one (B, total, ld) tensor
        total = sum(((H - 1) // stride + 1) * ((W - 1) // stride + 1) for H, W in levels) if pad != "same" else \
            sum((-(-H // stride)) * (-(-W // stride)) for H, W in levels)
        big = torch.full((B, total * ld + 8,), -77.0, dtype=odt, device=dev)
        keep.append(big)
    off = 0
    for gi, (H, W) in enumerate(levels):
        x = randn(B, H, W, cin, generator=g, dtype=torch.float64)
        if pad == "same":
            Ho, Wo = -(-H // stride), -(-W // stride)
            pt = max((Ho - 1) * stride + k - H, 0) // 2
            pl = max((Wo - 1) * stride + k - W, 0) // 2
        else:
            pt = pl = pad
            Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        d.pad_t, d.pad_l = pt, pl
        want = ref_conv(q(x, dtype), wq, bias.float().double(), stride, pt, pl, Ho, Wo) if reference else torch.zeros(B, Ho, Wo, cout, dtype=torch.float64)
        xd = x.to(tdt).to(dev).contiguous()
        grp = L.ConvGroup()
        grp.in_, grp.in_elems = xd.data_ptr(), xd.numel()
        grp.in_img_stride, grp.in_row_stride = H * W * cin, W * cin
        grp.Hin, grp.Win, grp.Hout, grp.Wout = H, W, Ho, Wo
        if res_mode == "same":
            r = randn(B, Ho, Wo, cout, generator=g, dtype=torch.float64)
            want = want + q(r, dtype)
        elif res_mode == "up":
            rh, rw = max(1, (Ho + 1) // 2), max(1, (Wo + 1) // 2 + (gi % 2))
            r = randn(B, rh, rw, cout, generator=g, dtype=torch.float64)
            want = want + upsample_nearest_legacy(q(r, dtype), Ho, Wo)
        if res_mode:
            rd = r.to(tdt).to(dev).contiguous()
            keep.append(rd)
            grp.res, grp.res_elems = rd.data_ptr(), rd.numel()
            grp.res_img_stride, grp.res_ld = rd.numel() // B, cout
            grp.Hres, grp.Wres = rd.shape[1], rd.shape[2]
        if flags & L.CONV_RELU:
            want = torch.relu(want)
        if flags & L.CONV_SIGMOID:
            want = torch.sigmoid(want)
        if concat:
            grp.out, grp.out_elems = big.data_ptr(), big.numel()
            grp.out_img_stride, grp.out_off = total * ld + 8, off
            outs.append((off, Ho * Wo))
            off += Ho * Wo * ld
        else:
            od = torch.full((B, Ho, Wo, ld), -77.0, dtype=odt, device=dev)
            keep.append(od)
            grp.out, grp.out_elems = od.data_ptr(), od.numel()
            grp.out_img_stride = Ho * Wo * ld
            outs.append(od)
        d.g[gi] = grp
        keep.append(xd)
        wants.append(want)
    d.flags = flags
    L.attach_conv_workspace(handle, d)            # the split-K / tail-split / stream-K paths take their slabs from the caller
    handle.check(L.lib.rtn_conv2d_fwd(handle.raw, C.byref(d)))
    torch.cuda.synchronize()
    LAST["d"], LAST["keep"] = d, keep
    gots = []
    for o, want in zip(outs, wants):
        if concat:
            off0, cells = o
            flat = big.cpu().double()
            gots.append(flat[:, off0:off0 + cells * ld].reshape(B, want.shape[1], want.shape[2], ld))
        else:
            gots.append(o.cpu().double())
    return gots, wants, ld, cout


def check(gots, wants, ld, cout, dtype):
    tol = 2e-5 if dtype == "f32" else 1e-2
    for got, want in zip(gots, wants):
        scale = max(1.0, float(want.abs().max()))
        err = float((got[..., :cout] - want).abs().max())
        assert err <= tol * scale, "max err %.3e (scale %.2f)" % (err, scale)
        if ld > cout:                                  # columns past N are never written
            assert torch.all(got[..., cout:] == -77.0)


CASES = [
    # (levels, cin, cout, k, stride, pad, flags-names, res)
    ([(9, 13)], 64, 64, 1, 1, 0, ["RELU"], None),                      # C2 branch2a-like, BN=64 tile
    ([(12, 20)], 64, 64, 3, 1, 1, ["RELU"], None),                     # 3x3, K step spans taps
    ([(17, 23)], 256, 128, 1, 2, 0, ["RELU"], None),                   # stride-2 1x1 'valid' (stage entry)
    ([(10, 11)], 128, 512, 1, 1, 0, ["RELU"], "same"),                 # branch2c + shortcut add
    ([(25, 42)], 256, 256, 3, 1, "same", [], None),                    # FPN P5
    ([(25, 42)], 512, 256, 3, 2, "same", [], None),                    # P6: stride 2, asymmetric TF pad
    ([(13, 21)], 256, 256, 3, 2, "same", [], None),                    # P7
    ([(20, 33)], 512, 256, 1, 1, 0, [], "up"),                         # lateral + UpsampleLike add (non-integer ratio)
    ([(16, 24), (8, 12), (4, 6), (2, 3), (1, 2)], 256, 256, 3, 1, "same", ["RELU"], None),   # grouped head layer
]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_conv_layers(pkg, handle, dtype, case):
    levels, cin, cout, k, stride, pad, fl, res = CASES[case]
    L = pkg._lib
    flags = sum(getattr(L, "CONV_" + f) for f in fl)
    if res == "same":
        flags |= L.CONV_RES_SAME
    elif res == "up":
        flags |= L.CONV_RES_UPSAMPLE
    gots, wants, ld, n = run_case(pkg, handle, dtype, levels, cin, cout, k, stride, pad, flags, res, seed=case)
    check(gots, wants, ld, n, dtype)


GEN_CASES = [
    # (levels, cin, cout, res, B): 3x3 stride-1 'same' layers, every kernel generation must agree with the reference
    ([(25, 42)], 256, 256, None, 3),                       # tiles of 254/256 rows straddle image rows and image boundaries
    ([(7, 5), (3, 2), (1, 1), (2, 1)], 128, 64, None, 2),  # grouped, widths 1 and 2: both horizontal edges on one pixel
    ([(31, 17)], 64, 192, "same", 2),                      # one 128-byte chunk per tap, N not a tile multiple, shortcut add
    ([(40, 67)], 256, 256, "up", 1),                       # upsampled residual; W = 67 never aligns with the tile
    ([(16, 24), (8, 12), (4, 6), (2, 3), (1, 2)], 256, 256, None, 2),
]


@pytest.mark.parametrize("impl", [1, 2, 3])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("case", range(len(GEN_CASES)))
def test_conv3x3_every_kernel_generation(pkg, handle, monkeypatch, dtype, case, impl):
    """RTN_CONV_IMPL pins the kernel generation (1: 128-row register-staged, 2: 256-row LDS-DMA per tap, 3: 256-row with
    the kernel row's taps sharing one staged halo; the library reads the knob on every call)."""
    levels, cin, cout, res, B = GEN_CASES[case]
    L = pkg._lib
    monkeypatch.setenv("RTN_CONV_IMPL", str(impl))
    flags = L.CONV_RELU | (L.CONV_RES_SAME if res == "same" else 0) | (L.CONV_RES_UPSAMPLE if res == "up" else 0)
    gots, wants, ld, n = run_case(pkg, handle, dtype, levels, cin, cout, 3, 1, "same", flags, res, B=B, seed=40 + case)
    check(gots, wants, ld, n, dtype)


@pytest.mark.parametrize("impl", [2, 3])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("levels,cin,cout,k,res,B,slots", [
    ([(25, 42)], 256, 256, 3, None, 3, 50),        # 13 row tiles x 4 column tiles (64 wide) = 52 tiles on 50 "slots": 2 tail tiles
    ([(31, 47)], 512, 192, 1, "same", 2, 34),      # 1x1, K = 8 steps, N not a tile multiple (36 tiles), shortcut add in the finish launch
    ([(40, 67)], 256, 256, 3, "up", 1, 42),        # 44 tiles
])
def test_tail_split(pkg, handle, monkeypatch, dtype, impl, levels, cin, cout, k, res, B, slots):
    """A grid a few tiles over a whole number of rounds gives its last tiles to K-slice workgroups of the same launch (f32
    slabs) and a finish launch that sums the slices in order and runs the epilogue.  RTN_CONV_TAIL_SLOTS pretends the chip
    has that many resident slots so that these small layers take that path; the result must still match the reference."""
    if impl == 3 and k == 1:
        pytest.skip("the halo kernel is for KW >= 2")
    L = pkg._lib
    monkeypatch.setenv("RTN_CONV_IMPL", str(impl))
    monkeypatch.setenv("RTN_CONV_BN2", "64")
    monkeypatch.setenv("RTN_CONV_TAIL_SLOTS", str(slots))
    flags = L.CONV_RELU | (L.CONV_RES_SAME if res == "same" else 0) | (L.CONV_RES_UPSAMPLE if res == "up" else 0)
    pad = "same" if k == 3 else 0
    gots, wants, ld, n = run_case(pkg, handle, dtype, levels, cin, cout, k, 1, pad, flags, res, B=B, seed=77)
    check(gots, wants, ld, n, dtype)
    monkeypatch.setenv("RTN_CONV_TAIL", "0")                      # and the same layer without the split
    gots0, _, _, _ = run_case(pkg, handle, dtype, levels, cin, cout, k, 1, pad, flags, res, B=B, seed=77)
    for a, b in zip(gots, gots0):
        assert float((a - b).abs().max()) <= (1e-5 if dtype == "f32" else 4e-2) * max(1.0, float(b.abs().max()))
        assert not torch.equal(a, b) or dtype == "bf16"      # f32: the split changes the summation order somewhere


@pytest.mark.parametrize("stagger,mi", [(1, 4), (0, 4), (1, 3), (0, 3)])
@pytest.mark.parametrize("levels,cin,cout,relu,B,grid", [
    ([(16, 24), (8, 12), (4, 6), (2, 3), (1, 2)], 256, 256, True, 2, 0),      # head-tower shape: five levels, one grouped launch
    ([(40, 67)], 256, 256, False, 3, 3),       # P3-like, no ReLU; 32 tiles on 3 workgroups: every workgroup walks ~11 tiles
    ([(25, 42), (13, 21)], 128, 200, True, 2, 1),   # two chunks per tap, N not a multiple of 16, ONE workgroup walks all 11 tiles
    ([(7, 300)], 256, 136, True, 1, 2),        # rows longer than a tile: tiles start and end inside an image row
    ([(3, 5)], 256, 256, True, 1, 0),          # a single, mostly empty tile
    ([(40, 67)], 128, 128, True, 3, 3),        # res3 branch2b: the 128-column instance (wave tile 64 columns, two phases per K step)
    ([(25, 42), (13, 21)], 256, 72, False, 2, 1),   # ... N = 72: second column half mostly empty, four chunks per tap, one workgroup
    ([(16, 24), (8, 12), (4, 6), (2, 3), (1, 2)], 128, 96, True, 2, 0),      # ... grouped levels
])
def test_persistent_8phase_halo_kernel(pkg, handle, monkeypatch, levels, cin, cout, relu, B, grid, stagger, mi):
    """Generation 4 (csrc/rtn_conv_halo8.hip): persistent 256 x 256 tiles on the staggered 8-phase schedule, LDS-DMA in flight
    across barriers behind counted waits, weight rows permuted for a register epilogue.  RTN_CONV_IMPL=4 takes it wherever it
    applies, RTN_CONV_H8_GRID limits the workgroup count so that workgroups walk several tiles (the halo / B-ring prefetch then
    crosses tile and pyramid-level boundaries), RTN_CONV_H8_STAGGER=0 runs the two wave groups in lockstep.  Against the float64
    convolution of the bf16-rounded operands; the launch must really have been generation 4."""
    L = pkg._lib
    monkeypatch.setenv("RTN_CONV_IMPL", "4")
    monkeypatch.setenv("RTN_CONV_H8_GRID", str(grid))
    monkeypatch.setenv("RTN_CONV_H8_STAGGER", str(stagger))
    monkeypatch.setenv("RTN_CONV_H8_MI", str(mi))            # tile height 64 * mi rows (256 / 192)
    flags = L.CONV_RELU if relu else 0
    gots, wants, ld, n = run_case(pkg, handle, "bf16", levels, cin, cout, 3, 1, "same", flags, None, B=B, seed=90 + grid)
    assert L.lib.rtn_debug_last_conv_impl(handle.raw) == 4
    check(gots, wants, ld, n, "bf16")
    # the same layer on generation 3: both are bf16 products summed in f32, in a different order
    monkeypatch.setenv("RTN_CONV_IMPL", "3")
    gots3, _, _, _ = run_case(pkg, handle, "bf16", levels, cin, cout, 3, 1, "same", flags, None, B=B, seed=90 + grid)
    assert L.lib.rtn_debug_last_conv_impl(handle.raw) == 3
    for a, b in zip(gots, gots3):
        assert float((a - b).abs().max()) <= 4e-2 * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("stagger", [1, 0])
@pytest.mark.parametrize("levels,cin,cout,stride,relu,B,grid,mi", [
    ([(25, 42)], 512, 256, 1, True, 8, 0, 0),      # res4 branch2a-like at batch 8: 8400 rows, 8 K steps, default grid and tile height
    ([(33, 50)], 256, 512, 2, True, 2, 3, 3),      # stride-2 'valid' 1x1 of a stage's first block, two N tiles, 3 workgroups walk 4 tiles each
    ([(17, 23)], 64, 256, 1, False, 3, 1, 2),      # ONE K step per tile: the rings turn over at every tile; one workgroup walks all tiles
    ([(40, 67)], 1024, 1024, 1, True, 1, 0, 0),    # 16 K steps, four N tiles
    ([(5, 7)], 128, 256, 1, True, 1, 0, 3),        # a single partly filled tile
])
def test_persistent_gemm8_kernel(pkg, handle, monkeypatch, levels, cin, cout, stride, relu, B, grid, mi, stagger):
    """Generation 5 (csrc/rtn_conv_gemm8.hip): the 1x1 layers with N % 256 == 0 as a persistent GEMM on the staggered 8-phase schedule
    (A ring of 3 issued last in every step, B ring of 2, counted waits, register epilogue).  RTN_CONV_IMPL=5 takes it wherever it
    applies; the launch must really have been generation 5.  Against the float64 product of the bf16-rounded operands, and against
    generation 2 on the same layer."""
    L = pkg._lib
    monkeypatch.setenv("RTN_CONV_IMPL", "5")
    monkeypatch.setenv("RTN_CONV_H8_GRID", str(grid))
    monkeypatch.setenv("RTN_CONV_H8_STAGGER", str(stagger))
    monkeypatch.setenv("RTN_CONV_G8_MI", str(mi))
    flags = L.CONV_RELU if relu else 0
    gots, wants, ld, n = run_case(pkg, handle, "bf16", levels, cin, cout, 1, stride, 0, flags, None, B=B, seed=60 + grid)
    assert L.lib.rtn_debug_last_conv_impl(handle.raw) == 5
    check(gots, wants, ld, n, "bf16")
    monkeypatch.setenv("RTN_CONV_IMPL", "2")
    gots2, _, _, _ = run_case(pkg, handle, "bf16", levels, cin, cout, 1, stride, 0, flags, None, B=B, seed=60 + grid)
    assert L.lib.rtn_debug_last_conv_impl(handle.raw) == 2
    for a, b in zip(gots, gots2):
        assert float((a - b).abs().max()) <= 4e-2 * max(1.0, float(b.abs().max()))


def sk_timeouts(pkg, handle):
    """Bounded flag polls of the last run_case's launch that gave up (the error word of its workspace's sync block)."""
    L = pkg._lib
    n = C.c_uint32(123)
    handle.check(L.lib.rtn_debug_conv_sync_timeouts(handle.raw, LAST["d"].workspace, C.byref(n)))
    return int(n.value)


@pytest.mark.parametrize("levels,cin,cout,stride,res,B,grid,mi", [
    ([(25, 42)], 2048, 512, 1, None, 2, 0, 0),     # res5 branch2a-like: 22 tiles x 32 K steps on 176 workgroups, 8 pieces per tile
    ([(25, 42)], 2048, 512, 1, None, 2, 7, 3),     # ... 7 workgroups: every range = tail of a tile + whole tiles + head of the next
    ([(25, 42)], 1024, 256, 1, "up", 3, 0, 3),     # C4_reduced-like: upsampled residual in the owner's epilogue
    ([(33, 50)], 128, 512, 1, "same", 2, 5, 2),    # branch2c + shortcut, two K steps per tile, 128-row tiles
    ([(33, 50)], 512, 512, 2, None, 2, 6, 2),      # stride-2 'valid' source
    ([(5, 7)], 1024, 256, 1, None, 1, 0, 3),       # ONE partly filled tile cut into 8 pieces of 2 K steps
    ([(40, 67)], 1024, 1024, 1, "same", 1, 0, 0),  # res4 branch2c-like: 4 column blocks, more tiles than a grid would need - the cost model's choice, forced
    ([(17, 23)], 256, 256, 1, None, 3, 3, 2),      # 4 K steps, 10 tiles on 3 workgroups: range boundaries at 13.3 and 26.7 steps
])
def test_gemm8_stream_k(pkg, handle, monkeypatch, levels, cin, cout, stride, res, B, grid, mi):
    """The stream-K form of generation 5 (RTN_CONV_G8_SK=1: wherever the shape allows): tiles x K steps cut into equal ranges, tails
    handed to the tile's owner through sc1 slabs + flags inside the launch.  (a) small-integer operands: BIT-IDENTICAL to the unsplit
    kernel and to the float64 reference (every sum is exact, so only a lost / doubled / misplaced piece can differ); (b) random
    operands: within the bf16 tolerance of float64 and of the unsplit kernel, and the same bits on every launch (fixed order of the
    adds); no poll timed out; the sync block is clean again (the second launch on the SAME workspace works)."""
    L = pkg._lib
    monkeypatch.setenv("RTN_CONV_IMPL", "5")
    monkeypatch.setenv("RTN_CONV_H8_GRID", str(grid))
    monkeypatch.setenv("RTN_CONV_G8_MI", str(mi))
    flags = (L.CONV_RES_SAME if res == "same" else 0) | (L.CONV_RES_UPSAMPLE if res == "up" else 0) | (0 if res == "up" else L.CONV_RELU)
    out = {}
    for sk in ("1", "0"):
        monkeypatch.setenv("RTN_CONV_G8_SK", sk)
        for exact in (True, False):
            gots, wants, ld, n = run_case(pkg, handle, "bf16", levels, cin, cout, 1, stride, 0, flags, res, B=B, seed=300 + grid, exact=exact)
            assert L.lib.rtn_debug_last_conv_impl(handle.raw) == 5
            assert (L.lib.rtn_debug_last_conv_streamk(handle.raw) > 0) == (sk == "1")
            if sk == "1":
                assert sk_timeouts(pkg, handle) == 0
                d = LAST["d"]                              # once more on the same workspace: the owner left the flags at zero
                handle.check(L.lib.rtn_conv2d_fwd(handle.raw, C.byref(d)))
                assert sk_timeouts(pkg, handle) == 0
            if exact:
                for a, b in zip(gots, wants):
                    assert torch.equal(a[..., :n], q(b, "bf16")), "exact operands: %.3e off the float64 result" % float((a[..., :n] - b).abs().max())
            else:
                check(gots, wants, ld, n, "bf16")
            out[(sk, exact)] = gots
    for a, b in zip(out[("1", True)], out[("0", True)]):
        assert torch.equal(a, b)
    for a, b in zip(out[("1", False)], out[("0", False)]):
        assert float((a - b).abs().max()) <= 8e-3 * max(1.0, float(b.abs().max()))           # one bf16 ulp where the f32 sums round apart
    monkeypatch.setenv("RTN_CONV_G8_SK", "1")
    for _ in range(2):
        again, _, _, _ = run_case(pkg, handle, "bf16", levels, cin, cout, 1, stride, 0, flags, res, B=B, seed=300 + grid, reference=False)
        for a, b in zip(again, out[("1", False)]):
            assert torch.equal(a, b)


@pytest.mark.parametrize("H,W,c1,c2,cout,step,B,grid", [
    (13, 21, 512, 1024, 2048, 2, 4, 0),        # res5a_branch2c + branch1: 24 K steps over two sources, 8 column blocks, the cost model's grid
    (25, 42, 256, 512, 1024, 2, 2, 5),         # res4a-like on 5 workgroups: ranges cross tiles AND the seam between the two sources
    (17, 23, 64, 64, 256, 1, 3, 3),            # one K step per source
])
def test_dual_source_stream_k(pkg, handle, monkeypatch, H, W, c1, c2, cout, step, B, grid):
    """rtn_conv1x1_dual_fwd (a stage's first block: branch2c with the projection shortcut appended along K, model/defineModel.py:376-380)
    in stream-K form: its workspace comes from rtn_conv1x1_dual_workspace_bytes.  Small-integer operands: bit-identical to the unsplit
    launch and to the float64 sum of the two products; random operands: same bits on every launch, within one bf16 ulp of the unsplit one."""
    L = pkg._lib
    dev = torch.device("cuda")
    monkeypatch.setenv("RTN_CONV_IMPL", "5")
    monkeypatch.setenv("RTN_CONV_H8_GRID", str(grid))
    g = torch.Generator().manual_seed(77 + grid)
    Hs, Ws = (H - 1) * step + 1, (W - 1) * step + 1

    def run(exact, sk):
        monkeypatch.setenv("RTN_CONV_G8_SK", sk)
        gg = torch.Generator().manual_seed(1000 + grid + (1 if exact else 0))
        if exact:
            a = torch.randint(-3, 4, (B, H, W, c1), generator=gg).double(); x2 = torch.randint(-3, 4, (B, Hs, Ws, c2), generator=gg).double()
            w = torch.randint(-2, 3, (cout, c1 + c2), generator=gg).double(); bias = torch.randint(-8, 9, (cout,), generator=gg).double()
        else:
            a = torch.randn(B, H, W, c1, generator=gg, dtype=torch.float64); x2 = torch.randn(B, Hs, Ws, c2, generator=gg, dtype=torch.float64)
            w = torch.randn(cout, c1 + c2, generator=gg, dtype=torch.float64) / math.sqrt(c1 + c2); bias = torch.randn(cout, generator=gg, dtype=torch.float64)
        aq, xq, wq = q(a, "bf16"), q(x2, "bf16"), q(w, "bf16")
        want = torch.relu(torch.einsum("bhwc,nc->bhwn", aq, wq[:, :c1]) + torch.einsum("bhwc,nc->bhwn", xq[:, ::step, ::step], wq[:, c1:]) + bias.float().double())
        ad, xd = a.to(torch.bfloat16).to(dev).contiguous(), x2.to(torch.bfloat16).to(dev).contiguous()
        wd, bd = w.to(torch.bfloat16).to(dev).contiguous(), bias.float().to(dev)
        out = torch.full((B, H, W, cout), -77.0, dtype=torch.bfloat16, device=dev)
        d = L.ConvDesc()
        d.ngroups, d.batch, d.dtype = 1, B, L.RTN_BF16
        d.w, d.bias, d.w_rows, d.N, d.KH, d.KW = wd.data_ptr(), bd.data_ptr(), cout, cout, 1, 1
        d.Crun = d.pix_stride = c1
        d.sy = d.sx = 1
        d.out_ld, d.flags = cout, L.CONV_RELU
        grp = d.g[0]
        grp.in_, grp.in_elems, grp.in_img_stride, grp.in_row_stride = ad.data_ptr(), ad.numel(), H * W * c1, W * c1
        grp.Hin, grp.Win, grp.Hout, grp.Wout = H, W, H, W
        grp.out, grp.out_elems, grp.out_img_stride = out.data_ptr(), out.numel(), H * W * cout
        s2 = L.ConvSrc2()
        s2.in_, s2.in_elems, s2.in_img_stride, s2.in_row_stride, s2.pix_stride = xd.data_ptr(), xd.numel(), Hs * Ws * c2, Ws * c2, c2
        s2.Hin, s2.Win, s2.C, s2.step = Hs, Ws, c2, step
        L.attach_conv_workspace(handle, d, s2)
        handle.check(L.lib.rtn_conv1x1_dual_fwd(handle.raw, C.byref(d), C.byref(s2)))
        torch.cuda.synchronize()
        assert L.lib.rtn_debug_last_conv_impl(handle.raw) == 5
        assert (L.lib.rtn_debug_last_conv_streamk(handle.raw) > 0) == (sk == "1")
        if sk == "1":
            n = C.c_uint32(9)
            handle.check(L.lib.rtn_debug_conv_sync_timeouts(handle.raw, d.workspace, C.byref(n)))
            assert n.value == 0
        return out.cpu().double(), want

    e1, want = run(True, "1")
    e0, _ = run(True, "0")
    assert torch.equal(e1, e0) and torch.equal(e1, q(want, "bf16"))
    r1, want = run(False, "1")
    r0, _ = run(False, "0")
    scale = max(1.0, float(want.abs().max()))
    assert float((r1 - want).abs().max()) <= 1e-2 * scale and float((r1 - r0).abs().max()) <= 8e-3 * scale
    r1b, _ = run(False, "1")
    assert torch.equal(r1, r1b)


@pytest.mark.parametrize("levels,cin,cout,B,grid,mi,ksplit", [
    ([(25, 42)], 512, 512, 2, 0, 0, 0),       # res5 branch2b: two column blocks; tile height and K slices by the cost model
    ([(25, 42)], 512, 512, 2, 3, 4, 1),       # ... unsliced: the epilogue adds the bias (accumulators of a block item start at zero)
    ([(25, 42)], 512, 512, 1, 5, 3, 3),       # ... 3 slices (one kernel row each), 5 workgroups walk ~36 items each
    ([(25, 42)], 256, 320, 2, 2, 4, 6),       # second column block a quarter full, weight rows past 384 come from the range check
    ([(13, 21)], 256, 256, 3, 0, 0, 4),       # one column block, 4 slices of 3 groups: P5-like
    ([(40, 67)], 128, 256, 1, 7, 3, 2),       # two chunks per tap: slices of 3 groups start in the middle of a kernel row
])
def test_halo8_column_blocks_and_k_slices(pkg, handle, monkeypatch, levels, cin, cout, B, grid, mi, ksplit):
    """Generation 4 on the small-M, long-K layers: work items = (row tile, column block of 256 channels, K slice); the slices' f32
    partial sums go to caller-owned slabs (rtn_conv2d_workspace_bytes) and are added in slice order by ksplit_finish_kernel, so
    repeated launches give the same bits.  RTN_CONV_H8_KSPLIT pins the slice count (1 = never)."""
    L = pkg._lib
    monkeypatch.setenv("RTN_CONV_IMPL", "4")
    monkeypatch.setenv("RTN_CONV_H8_GRID", str(grid))
    monkeypatch.setenv("RTN_CONV_H8_MI", str(mi))
    monkeypatch.setenv("RTN_CONV_H8_KSPLIT", str(ksplit))
    gots, wants, ld, n = run_case(pkg, handle, "bf16", levels, cin, cout, 3, 1, "same", L.CONV_RELU, None, B=B, seed=120 + grid)
    assert L.lib.rtn_debug_last_conv_impl(handle.raw) == 4
    check(gots, wants, ld, n, "bf16")
    for _ in range(3):
        again, _, _, _ = run_case(pkg, handle, "bf16", levels, cin, cout, 3, 1, "same", L.CONV_RELU, None, B=B, seed=120 + grid, reference=False)
        for a, b in zip(again, gots):
            assert torch.equal(a, b)


@pytest.mark.parametrize("levels,cin,cout,res,B,grid,mi", [
    ([(40, 67)], 512, 256, "up", 2, 0, 0),         # C3_reduced-like: lateral 1x1 + UpsampleLike(coarser level) + Add, non-integer ratio
    ([(25, 42)], 1024, 256, "up", 3, 2, 2),        # C4_reduced-like on 2 workgroups
    ([(33, 50)], 128, 512, "same", 2, 3, 3),       # branch2c + identity shortcut
    ([(40, 67)], 512, 128, None, 2, 0, 0),         # res3 branch2a: N = 128, the tile's upper 128 columns multiply zeros
    ([(17, 23)], 64, 128, "same", 3, 1, 2),
])
def test_gemm8_residual_forms_and_half_width(pkg, handle, monkeypatch, levels, cin, cout, res, B, grid, mi):
    """Generation 5 epilogues: identity shortcut, the FPN lateral's upsampled residual (model/layers.py:89-98 through
    RTN_CONV_RES_UPSAMPLE), and N = 128 layers on the 256-column tile."""
    L = pkg._lib
    monkeypatch.setenv("RTN_CONV_IMPL", "5")
    monkeypatch.setenv("RTN_CONV_H8_GRID", str(grid))
    monkeypatch.setenv("RTN_CONV_G8_MI", str(mi))
    flags = (L.CONV_RES_SAME if res == "same" else 0) | (L.CONV_RES_UPSAMPLE if res == "up" else 0) | (0 if res == "up" else L.CONV_RELU)
    gots, wants, ld, n = run_case(pkg, handle, "bf16", levels, cin, cout, 1, 1, 0, flags, res, B=B, seed=140 + grid)
    assert L.lib.rtn_debug_last_conv_impl(handle.raw) == 5
    check(gots, wants, ld, n, "bf16")


@pytest.mark.parametrize("levels,cin,stride,res,B,grid,mi", [
    ([(40, 67)], 512, 1, None, 2, 0, 0),           # res3 branch2a: 8 K steps, default grid and tile height
    ([(40, 67)], 512, 1, None, 2, 3, 2),           # ... 3 workgroups walk 14 tiles of 128 rows each: ring turnover across tiles
    ([(33, 51)], 256, 2, None, 3, 2, 3),           # res3a branch2a: stride-2 'valid' sampling, 4 K steps
    ([(17, 23)], 64, 1, "same", 3, 1, 2),          # ONE K step per tile (the B ring runs two tiles ahead), residual epilogue (8-byte loads)
    ([(25, 42)], 1024, 1, "same", 2, 5, 3),        # 16 K steps, residual
])
def test_gemm8_narrow_instance_for_128_columns(pkg, handle, monkeypatch, levels, cin, stride, res, B, grid, mi):
    """conv_gemm8_kernel<.., NW = 4>: the 128-column tile for the N = 128 layers (res3 branch2a behind model/defineModel.py:376-380 and
    the matching data gradients): two phases per K step, B ring of three 16 KiB stages, 8-byte register stores.  Against the float64
    product of the bf16 operands, BIT FOR BIT against the same layer on the 256-column tile (RTN_CONV_G8_NARROW=0: the same k order and
    the same bias-initialised accumulators), and bit for bit against itself on repeated launches."""
    L = pkg._lib
    monkeypatch.setenv("RTN_CONV_IMPL", "5")
    monkeypatch.setenv("RTN_CONV_H8_GRID", str(grid))
    monkeypatch.setenv("RTN_CONV_G8_MI", str(mi))
    flags = L.CONV_RELU | (L.CONV_RES_SAME if res == "same" else 0)
    out = {}
    for narrow in ("1", "0", "1"):
        monkeypatch.setenv("RTN_CONV_G8_NARROW", narrow)
        gots, wants, ld, n = run_case(pkg, handle, "bf16", levels, cin, 128, 1, stride, 0, flags, res, B=B, seed=170 + grid)
        assert L.lib.rtn_debug_last_conv_impl(handle.raw) == 5
        check(gots, wants, ld, n, "bf16")
        out.setdefault(narrow, []).append(gots)
    for a, b in zip(out["1"][0], out["1"][1]):
        assert torch.equal(a, b)
    for a, b in zip(out["1"][0], out["0"][0]):
        assert torch.equal(a, b), "narrow and wide instances differ by %.3e" % float((a - b).abs().max())



def test_persistent_8phase_kernel_repeats_bit_for_bit(pkg, handle, monkeypatch):
    """A race between an LDS-DMA piece and a fragment read shows up as a tile that changes from launch to launch: 6 launches of a
    head-tower-sized layer (five levels, 700 tiles, every CU walking 2-3 of them) must give the same bits, staggered and not."""
    L = pkg._lib
    monkeypatch.setenv("RTN_CONV_IMPL", "4")
    levels = [(100, 167), (50, 84), (25, 42), (13, 21), (7, 11)]
    first = None
    for it in range(6):
        monkeypatch.setenv("RTN_CONV_H8_STAGGER", str(1 - it % 2))
        monkeypatch.setenv("RTN_CONV_H8_MI", "4" if it < 4 else "3")
        gots, wants, ld, n = run_case(pkg, handle, "bf16", levels, 256, 256, 3, 1, "same", L.CONV_RELU, None, B=8, seed=5,
                                      reference=(it == 0))
        assert L.lib.rtn_debug_last_conv_impl(handle.raw) == 4
        if first is None:
            check(gots, wants, ld, n, "bf16")
            first = gots
        else:
            for a, b in zip(gots, first):           # the tile height does not change a pixel's summation order either
                assert torch.equal(a, b), "launch %d differs" % it


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("cout,sig", [(36, False), (9, True), (18, True)])
def test_head_output_concat(pkg, handle, dtype, cout, sig):
    """pyramid_regression / pyramid_classification: skinny N, f32 output written at level offsets of the
    concatenated (B, N_anchors, 4|K) tensor (model/defineModel.py:111-123,163-166,217)."""
    L = pkg._lib
    flags = L.CONV_OUT_F32 | (L.CONV_SIGMOID if sig else 0)
    levels = [(16, 24), (8, 12), (4, 6), (2, 3), (1, 2)]
    gots, wants, ld, n = run_case(pkg, handle, dtype, levels, 256, cout, 3, 1, "same", flags, None, concat=True, seed=7)
    check(gots, wants, ld, n, dtype)


@pytest.mark.parametrize("levels,cin,cout,sig,B,grid", [
    ([(16, 24), (8, 12), (4, 6), (2, 3), (1, 2)], 256, 36, False, 2, 0),     # pyramid_regression: 16-byte f32 stores
    ([(16, 24), (8, 12), (4, 6), (2, 3), (1, 2)], 256, 9, True, 2, 0),       # pyramid_classification: 9 channels, dword stores, sigmoid
    ([(40, 67), (20, 34)], 256, 36, False, 3, 3),      # 3 workgroups walk ~12 tiles each: the stage ring crosses tiles and levels
    ([(25, 42), (13, 21)], 128, 18, True, 2, 1),       # two anchors' worth of channels (2 fragments), two chunks per tap, ONE workgroup
    ([(7, 300)], 64, 48, False, 1, 2),                 # rows longer than a tile, one chunk per tap, 3 full fragments
    ([(3, 5)], 256, 4, False, 1, 0),                   # a single, mostly empty tile
])
def test_persistent_head_output_kernel(pkg, handle, monkeypatch, levels, cin, cout, sig, B, grid):
    """Generation 6 (csrc/rtn_conv_halon.hip): the head output convolutions (3x3, <= 48 channels, f32 result at the level offsets
    of the concatenated tensor) as a persistent kernel with transposed products.  Against the float64 convolution of the
    bf16-rounded operands; the launch must really have been generation 6; repeated launches give the same bits."""
    L = pkg._lib
    monkeypatch.setenv("RTN_CONV_IMPL", "6")
    monkeypatch.setenv("RTN_CONV_H8_GRID", str(grid))
    flags = L.CONV_OUT_F32 | (L.CONV_SIGMOID if sig else 0)
    gots, wants, ld, n = run_case(pkg, handle, "bf16", levels, cin, cout, 3, 1, "same", flags, None, B=B, concat=True, seed=30 + grid)
    assert L.lib.rtn_debug_last_conv_impl(handle.raw) == 6
    check(gots, wants, ld, n, "bf16")
    for _ in range(3):
        again, _, _, _ = run_case(pkg, handle, "bf16", levels, cin, cout, 3, 1, "same", flags, None, B=B, concat=True, seed=30 + grid,
                                  reference=False)
        for a, b in zip(again, gots):
            assert torch.equal(a, b)


def test_conv_rejects_bad_descriptors(pkg, handle):
    L = pkg._lib
    dev = torch.device("cuda")
    x = torch.zeros(1, 4, 4, 64, dtype=torch.bfloat16, device=dev)
    w = torch.zeros(128, 64, dtype=torch.bfloat16, device=dev)
    o = torch.zeros(1, 4, 4, 64, dtype=torch.bfloat16, device=dev)
    d = L.ConvDesc()
    d.ngroups, d.batch, d.dtype = 1, 1, 0
    d.w, d.w_rows, d.N, d.KH, d.KW, d.Crun, d.pix_stride = w.data_ptr(), 128, 64, 1, 1, 64, 64
    d.sy = d.sx = 1
    d.out_ld = 64
    g = L.ConvGroup()
    g.in_, g.in_elems, g.out, g.out_elems = x.data_ptr(), x.numel(), o.data_ptr(), o.numel()
    g.in_img_stride, g.in_row_stride, g.out_img_stride = 4 * 4 * 64, 4 * 64, 4 * 4 * 64
    g.Hin = g.Win = g.Hout = g.Wout = 4
    d.g[0] = g
    assert L.lib.rtn_conv2d_fwd(handle.raw, C.byref(d)) == 0
    d.g[0].in_elems = x.numel() - 1                       # taps would leave the buffer
    assert L.lib.rtn_conv2d_fwd(handle.raw, C.byref(d)) == -4
    assert b"taps reach" in L.lib.rtn_last_error(handle.raw)
    d.g[0].in_elems = x.numel()
    d.g[0].out_elems = o.numel() - 1
    assert L.lib.rtn_conv2d_fwd(handle.raw, C.byref(d)) == -4
    d.g[0].out_elems = o.numel()
    d.Crun = 48                                           # not a power of two
    assert L.lib.rtn_conv2d_fwd(handle.raw, C.byref(d)) == -1
    d.Crun = 64
    d.w_rows = 64                                         # not padded to 128 rows
    assert L.lib.rtn_conv2d_fwd(handle.raw, C.byref(d)) == -1
    torch.cuda.synchronize()


def test_workspace_query_never_launches(pkg, handle, monkeypatch):
    """rtn_conv2d_workspace_bytes is conv_launch in query mode.  With RTN_CONV_SPLITK=0 the split-K sizing block (and its early
    return) is skipped for a tiny-M long-K layer on generation 1; the query must still return before any kernel is launched: the
    output buffer keeps its fill value and the handle's last-kernel record does not change."""
    L = pkg._lib
    dev = torch.device("cuda")
    B, H, W, cin, cout = 1, 6, 7, 2048, 64
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, H, W, cin, generator=g).to(torch.bfloat16).to(dev)
    wk, rows = pack_w(torch.randn(1, 1, cin, cout, generator=g, dtype=torch.float64) / 45.0, "bf16", dev)
    bk = torch.zeros(rows, dtype=torch.float32, device=dev)
    out = torch.full((B, H, W, cout), -77.0, dtype=torch.bfloat16, device=dev)
    d = L.ConvDesc()
    d.ngroups, d.batch, d.dtype = 1, B, 0
    d.w, d.bias, d.w_rows, d.N, d.KH, d.KW = wk.data_ptr(), bk.data_ptr(), rows, cout, 1, 1
    d.Crun = d.pix_stride = cin
    d.sy = d.sx = 1
    d.out_ld = cout
    grp = L.ConvGroup()
    grp.in_, grp.in_elems, grp.in_img_stride, grp.in_row_stride = x.data_ptr(), x.numel(), H * W * cin, W * cin
    grp.Hin, grp.Win, grp.Hout, grp.Wout = H, W, H, W
    grp.out, grp.out_elems, grp.out_img_stride = out.data_ptr(), out.numel(), H * W * cout
    d.g[0] = grp
    monkeypatch.setenv("RTN_CONV_IMPL", "1")
    for splitk in ("0", "4"):
        monkeypatch.setenv("RTN_CONV_SPLITK", splitk)
        before = L.lib.rtn_debug_last_conv_impl(handle.raw)
        nbytes = L.lib.rtn_conv2d_workspace_bytes(handle.raw, C.byref(d))
        torch.cuda.synchronize()
        assert (nbytes > 0) == (splitk == "4")
        assert torch.all(out == -77.0), "the workspace query wrote the output (RTN_CONV_SPLITK=%s)" % splitk
        assert L.lib.rtn_debug_last_conv_impl(handle.raw) == before
    # and the real launch still works with the split off
    monkeypatch.setenv("RTN_CONV_SPLITK", "0")
    handle.check(L.lib.rtn_conv2d_fwd(handle.raw, C.byref(d)))
    torch.cuda.synchronize()
    want = (x.double().reshape(-1, cin) @ wk[:cout].double().t().cpu().to(dev)).reshape(B, H, W, cout)
    assert float((out.double() - want).abs().max()) <= 1e-2 * max(1.0, float(want.abs().max()))
