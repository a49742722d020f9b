"""Time one head-tower layer (3x3, 256 -> 256, ReLU, the five pyramid levels grouped) in bf16 (rtn_conv2d_fwd) and in fp8
(rtn_conv2d_fp8_fwd, fp8 output):  python tools/bench_fp8_conv.py [--canvas 800 1333] [--batch 8] [--iters 30]"""
import argparse
import ctypes as C
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("retinanet-for-table-detection_amd")
L = pkg._lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--canvas", type=int, nargs=2, default=[800, 1333])
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    H, W = a.canvas
    levels = [(-(-H // 2 ** l), -(-W // 2 ** l)) for l in (3, 4, 5, 6, 7)]
    h = pkg.Handle(0)
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    B, Cc = a.batch, 256
    res = {}
    for name, es in (("bf16", 2), ("fp8", 1)):
        tdt = torch.bfloat16 if es == 2 else torch.uint8
        w = (torch.randn(256, 9 * Cc, device="cuda") * 0.02).to(torch.bfloat16)
        if es == 1:
            w = torch.clamp(w.float() * 4000, -448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
        bias = torch.zeros(256, device="cuda")
        d = L.ConvDesc()
        d.ngroups, d.batch, d.dtype = 5, B, (L.RTN_BF16 if es == 2 else L.RTN_FP8)
        d.w, d.bias, d.w_rows, d.N, d.KH, d.KW = w.data_ptr(), bias.data_ptr(), 256, 256, 3, 3
        d.Crun = d.pix_stride = Cc
        d.sy = d.sx = d.pad_t = d.pad_l = 1
        d.out_ld = Cc
        d.flags = L.CONV_RELU
        keep = []
        flop = 0
        for gi, (hh, ww) in enumerate(levels):
            x = torch.randn(B, hh, ww, Cc, device="cuda")
            x = x.to(torch.bfloat16) if es == 2 else torch.clamp(x * 100, -448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
            o = torch.empty(B, hh, ww, Cc, dtype=tdt, device="cuda")
            keep += [x, o]
            g = d.g[gi]
            g.in_, g.in_elems, g.in_img_stride, g.in_row_stride = x.data_ptr(), x.numel(), hh * ww * Cc, ww * Cc
            g.Hin, g.Win, g.Hout, g.Wout = hh, ww, hh, ww
            g.out, g.out_elems, g.out_img_stride = o.data_ptr(), o.numel(), hh * ww * Cc
            flop += 2 * B * hh * ww * 9 * Cc * 256
        q = L.ConvFp8(acc_scale=1e-5, out_scale=1.0, out_dtype=L.RTN_FP8)

        def run():
            if es == 2:
                h.check(pkg.lib.rtn_conv2d_fwd(h.raw, C.byref(d)))
            else:
                h.check(pkg.lib.rtn_conv2d_fp8_fwd(h.raw, C.byref(d), C.byref(q)))
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        res[name] = ms
        print("%-5s %.3f ms per layer launch, %.0f TFLOP/s (%.1f GFLOP), kernel generation %d" %
              (name, ms, flop / ms / 1e9, flop / 1e9, pkg.lib.rtn_debug_last_conv_impl(h.raw)))
    print("fp8 / bf16 time: %.2f" % (res["fp8"] / res["bf16"]))


if __name__ == "__main__":
    main()
