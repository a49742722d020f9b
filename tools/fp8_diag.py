"""Where the fp8 convolution differs from the float64 reference of the same e4m3 bytes: the elements more than one code apart
(all near zero) and the accumulate noise floor in units of the output scale (cited by tests/test_gpu_fp8.py)."""
import importlib, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_fp8 as T
pkg = importlib.import_module("retinanet-for-table-detection_amd")
h = pkg.Handle(0); h.set_stream(torch.cuda.current_stream().cuda_stream)
outs, wants, so = T.run_fp8(pkg, h, [(25, 42), (13, 21), (7, 11)], 256, 256, 3, True, out_fp8=True, seed=3)
for o, w in zip(outs, wants):
    want = T.to_f8(w.float(), so).view(torch.uint8)
    d = T.code_distance(o.cpu(), want)
    idx = (d > 1).nonzero()
    print("shape", tuple(o.shape), "mismatch frac", float((d > 0).float().mean()), "n>1:", len(idx))
    for i in idx[:6]:
        i = tuple(int(v) for v in i)
        print("  ref y*so = %.6f  want code %d  got code %d" % (float(w[i]) * so, int(want[i]), int(o.cpu()[i])))
# bf16-output error in units of the output scale, to see the accumulate noise floor
outs, wants, _ = T.run_fp8(pkg, h, [(25, 42)], 256, 256, 3, False, out_fp8=False, seed=3)
g = outs[0].double().cpu(); w = wants[0]
err = (g - w).abs()
print("bf16 out: max err / scale = %.3e ; err on |w|<0.01*scale: %.3e" % (float(err.max() / w.abs().max()), float(err[w.abs() < 0.01 * w.abs().max()].max() / w.abs().max())))
