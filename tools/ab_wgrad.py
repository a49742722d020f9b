"""Same-process A/B of the weight-gradient kernels on one layer shape (head tower by default): interleaved rounds, median.
   python tools/ab_wgrad.py [tower|p3|res4|res3|res5] VARIANT[,VARIANT...]   with VARIANT = ENV=VAL[+ENV=VAL...]"""
import ctypes as C, importlib, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, bench
L = importlib.import_module(bench.PKG + "._lib")
SHAPES = {"tower": ([(100, 167), (50, 84), (25, 42), (13, 21), (7, 11)], 256, 256), "p3": ([(100, 167)], 256, 256),
          "res4": ([(50, 84)], 256, 256), "res3": ([(100, 167)], 128, 128), "res5": ([(25, 42)], 512, 512), "p5": ([(25, 42)], 256, 256), "res2": ([(200, 334)], 64, 64), "headout": ([(100, 167), (50, 84), (25, 42), (13, 21), (7, 11)], 256, 64), "res4x2": ([(50, 84)], 256, 256),
          # 1x1 layers (cin, cout, k = 1)
          "res4_2a": ([(50, 84)], 1024, 256, 1), "res4_2c": ([(50, 84)], 256, 1024, 1), "res3_2a": ([(100, 167)], 512, 128, 1),
          "res3_2c": ([(100, 167)], 128, 512, 1), "res5_2a": ([(25, 42)], 2048, 512, 1), "res5_2c": ([(25, 42)], 512, 2048, 1),
          "res2_2a": ([(200, 334)], 256, 64, 1), "res2_2c": ([(200, 334)], 64, 256, 1), "c3": ([(100, 167)], 512, 256, 1)}
name = sys.argv[1] if len(sys.argv) > 1 else "tower"
variants = [dict(kv.split("=") for kv in v.split("+")) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["RTN_WGRAD_WIN=1", "RTN_WGRAD_WIN=0"])]
levels, cin, cout = SHAPES[name][:3]
K = SHAPES[name][3] if len(SHAPES[name]) > 3 else 3
B = int(os.environ.get("AB_BATCH", bench.BATCH))
h = L.Handle(0)
d = L.ConvDesc()
d.ngroups, d.batch, d.dtype = len(levels), B, 0
d.w_rows, d.N, d.KH, d.KW = cout, cout, K, K
d.Crun = d.pix_stride = cin
d.sy = d.sx = 1; d.pad_t = d.pad_l = K // 2; d.out_ld = cout
keep = []
for gi, (H, W) in enumerate(levels):
    x = torch.randn(B, H, W, cin, device="cuda").to(torch.bfloat16); dy = torch.randn(B, H, W, cout, device="cuda").to(torch.bfloat16)
    keep += [x, dy]
    g = L.ConvGroup()
    g.in_, g.in_elems, g.in_img_stride, g.in_row_stride = x.data_ptr(), x.numel(), H * W * cin, W * cin
    g.Hin, g.Win, g.Hout, g.Wout = H, W, H, W
    g.out, g.out_elems, g.out_img_stride = dy.data_ptr(), dy.numel(), H * W * cout
    d.g[gi] = g
flops = 2.0 * sum(H * W for H, W in levels) * B * cout * K * K * cin
KNOBS = sorted({k for v in variants for k in v})
dW = torch.zeros(cout, K * K * cin, device="cuda"); db = torch.zeros(cout, device="cuda")
times = {i: [] for i in range(len(variants))}
for rnd in range(10):
    for i, v in enumerate(variants):
        for k in KNOBS: os.environ.pop(k, None)
        os.environ.update(v)
        wsb = L.lib.rtn_conv2d_wgrad_workspace_bytes(C.byref(d)); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        # as the training step does: the row-info table once (rtn_conv2d_wgrad_rowinfo), then prepared launches - the general kernels' A/B
        # baseline must not carry a 16 B-per-pixel table write that the window kernel's does not (round-3 advisor note)
        h.check(L.lib.rtn_conv2d_wgrad_rowinfo(h.raw, C.byref(d), ws.data_ptr(), wsb))
        h.check(L.lib.rtn_conv2d_wgrad_prepared(h.raw, C.byref(d), dW.data_ptr(), db.data_ptr(), cout, ws.data_ptr(), wsb)); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): h.check(L.lib.rtn_conv2d_wgrad_prepared(h.raw, C.byref(d), dW.data_ptr(), db.data_ptr(), cout, ws.data_ptr(), wsb))
        e.record(); torch.cuda.synchronize()
        if rnd >= 2: times[i].append(s.elapsed_time(e) / 5)
for i, v in enumerate(variants):
    med = statistics.median(times[i])
    print("%-8s %-44s median %.4f ms (%.0f TF/s)" % (name, "+".join("%s=%s" % kv for kv in v.items()), med, flops / med / 1e9))
