"""Experiment: is the 256 x 256 weight-gradient loop bound by HBM-miss latency behind in-order vmcnt?  The same launch on B images that
are DISTINCT in memory vs B images that alias ONE image (image stride 0: the whole pixel stream stays L2-resident after the first
image).  Same instruction stream, same L2 -> LDS volume; only the miss rate differs."""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, bench
L = importlib.import_module(bench.PKG + "._lib")
os.environ["RTN_WGRAD_DMA"] = "2"
H, W, cin, cout, B = 25, 42, 256, 256, 128      # one image = 1.07 MB of X + dY: resident in every XCD's 4 MB L2 when aliased
h = L.Handle(0)
for alias in (0, 1, 0, 1):
    x = torch.randn(B if not alias else 1, H, W, cin, device="cuda").to(torch.bfloat16)
    dy = torch.randn(B if not alias else 1, H, W, cout, device="cuda").to(torch.bfloat16)
    d = L.ConvDesc()
    d.ngroups, d.batch, d.dtype = 1, B, 0
    d.w_rows, d.N, d.KH, d.KW = cout, cout, 3, 3
    d.Crun = d.pix_stride = cin
    d.sy = d.sx = 1; d.pad_t = d.pad_l = 1; d.out_ld = cout
    g = L.ConvGroup()
    g.in_, g.in_elems, g.in_img_stride, g.in_row_stride = x.data_ptr(), x.numel(), (0 if alias else H * W * cin), W * cin
    g.Hin, g.Win, g.Hout, g.Wout = H, W, H, W
    g.out, g.out_elems, g.out_img_stride = dy.data_ptr(), dy.numel(), (0 if alias else H * W * cout)
    d.g[0] = g
    dW = torch.zeros(cout, 9 * cin, device="cuda"); db = torch.zeros(cout, device="cuda")
    wsb = L.lib.rtn_conv2d_wgrad_workspace_bytes(C.byref(d)); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    h.check(L.lib.rtn_conv2d_wgrad_rowinfo(h.raw, C.byref(d), ws.data_ptr(), wsb))
    for ring in ("0", "1"):
        os.environ["RTN_WGRAD_RING"] = ring
        for _ in range(3):
            h.check(L.lib.rtn_conv2d_wgrad_prepared(h.raw, C.byref(d), dW.data_ptr(), db.data_ptr(), cout, ws.data_ptr(), wsb))
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            h.check(L.lib.rtn_conv2d_wgrad_prepared(h.raw, C.byref(d), dW.data_ptr(), db.data_ptr(), cout, ws.data_ptr(), wsb))
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 10
        fl = 2.0 * B * H * W * cout * 9 * cin
        print("images %s, ring=%s: %.4f ms per launch incl. the finish kernel (%.0f TF/s)" % ("ALIASED (L2-resident stream)" if alias else "distinct", ring, ms, fl / ms / 1e9))
