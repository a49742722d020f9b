"""Timing of the K20 preprocessing kernels on a batch of 2200x1712 pages (the reference sample page size)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
L = importlib.import_module(bench.PKG + "._lib")
B, H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 2200, 1712
h = L.Handle(0); h.set_stream(torch.cuda.current_stream().cuda_stream)
rng = np.random.RandomState(0)
page = np.full((H, W), 250, np.uint8)
for _ in range(4000):                       # synthetic "text": small dark strokes
    y, x = rng.randint(0, H - 4), rng.randint(0, W - 30)
    page[y:y + 3, x:x + rng.randint(5, 30)] = rng.randint(0, 90)
src = torch.as_tensor(np.stack([page] * B)).cuda()
dst = torch.empty(B, H, W, 3, dtype=torch.uint8, device="cuda")
wsb = L.lib.rtn_preprocess_dt3_workspace_bytes(B, H, W); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
def run(): h.check(L.lib.rtn_preprocess_dt3(h.raw, src.data_ptr(), 1, B, H, W, dst.data_ptr(), None, ws.data_ptr(), wsb))
run(); torch.cuda.synchronize()
t0 = time.perf_counter(); n = 5
for _ in range(n): run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print("preprocess_dt3: %d pages %dx%d in %.2f ms -> %.0f pages/s" % (B, H, W, dt * 1e3, B / dt))
scale = 800.0 / W
Ho, Wo = int(np.rint(H * scale)), int(np.rint(W * scale))
canvas = torch.zeros(B, Ho, Wo, 3, dtype=torch.bfloat16, device="cuda")
def rs():
    for i in range(B): h.check(L.lib.rtn_resize_cubic(h.raw, dst[i].data_ptr(), 2, H, W, 3, scale, canvas[i].data_ptr(), 0, Ho, Wo, Wo * 3))
rs(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): rs()
torch.cuda.synchronize(); dt2 = (time.perf_counter() - t0) / n
print("normalise + resize_cubic into canvas: %.2f ms for %d pages -> %.0f pages/s" % (dt2 * 1e3, B, B / dt2))
