"""Same-process A/B of the fused branch2c -> next branch2a kernel (rtn_chain1x1_fwd) against the two rtn_conv2d_fwd launches it
replaces, at the bench's stage-3 / stage-4 sizes: interleaved rounds, median of the event times.   python tools/ab_chain.py [res3] [batch] [VARIANTS]"""
import ctypes as C, importlib, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench
L = importlib.import_module(bench.PKG + "._lib")
from test_gpu_chain import conv1x1
which = sys.argv[1] if len(sys.argv) > 1 else "res3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else bench.BATCH
mid, (H, W) = {"res3": (128, (100, 167))}[which]      # (the 256-channel instance left the library: profiles/r4_seam_kernel.txt)
M, out = B * H * W, 4 * mid
h = L.Handle(0)
dev = torch.device("cuda")
r16 = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
hd, xd = torch.relu(r16(M, mid)), torch.relu(r16(M, out))
wc, wa = (r16(out, mid) / mid ** 0.5).to(torch.bfloat16), (r16(mid, out) / out ** 0.5).to(torch.bfloat16)
bc, ba = torch.randn(out, device=dev) * 0.3, torch.randn(mid, device=dev) * 0.3
xo, ao, x2, a2 = [torch.empty(M, n, dtype=torch.bfloat16, device=dev) for n in (out, mid, out, mid)]
d = L.ChainDesc()
d.h_in, d.h_in_elems, d.x_in, d.x_in_elems = hd.data_ptr(), hd.numel(), xd.data_ptr(), xd.numel()
d.x_out, d.x_out_elems, d.a_out, d.a_out_elems = xo.data_ptr(), xo.numel(), ao.data_ptr(), ao.numel()
d.w2c, d.b2c, d.w2a, d.b2a = wc.data_ptr(), bc.data_ptr(), wa.data_ptr(), ba.data_ptr()
d.pixels, d.mid, d.out, d.next, d.dtype = M, mid, out, mid, L.RTN_BF16
keep = []
def separate():
    keep.append(conv1x1(L, h, hd, x2, wc, bc, out, mid, L.CONV_RELU | L.CONV_RES_SAME, res=xd))
    keep.append(conv1x1(L, h, x2, a2, wa, ba, mid, out, L.CONV_RELU))
def fused(**env):
    def run():
        os.environ.update(env)
        h.check(L.lib.rtn_chain1x1_fwd(h.raw, C.byref(d)))
    return run
VARIANTS = [dict(kv.split("=") for kv in v.split("+")) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["RTN_CHAIN_SPREAD=1", "RTN_CHAIN_SPREAD=0"])]
cases = [("separate", separate)] + [("fused " + " ".join("%s=%s" % kv for kv in v.items()), fused(**v)) for v in VARIANTS]
times = {name: [] for name, _ in cases}
for rnd in range(12):
    for name, fn in cases:
        fn(); torch.cuda.synchronize(); del keep[:]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize(); del keep[:]
        if rnd >= 2: times[name].append(e0.elapsed_time(e1) / 20)
print("%s batch %d: M = %d, %d -> %d -> %d; equal bits: x_out %s, a_out %s" % (which, B, M, mid, out, mid, torch.equal(xo, x2), torch.equal(ao, a2)))
flops = 2.0 * M * (mid * out * 2)
byts = 2.0 * M * (mid + out + out + mid)
for name, ts in times.items():
    t = statistics.median(ts)
    print("  %-28s %.4f ms  (%.0f TF/s; fused-form bytes %.0f MB -> %.2f TB/s)" % (name, t, flops / t / 1e9, byts / 1e6, byts / t / 1e9))
