"""Time rtn_filter_detections (explicit boxes) on synthetic candidate sets: disjoint, clustered, heavy overlap."""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
L = importlib.import_module("retinanet-for-table-detection_amd._lib")

def make(kind, B, N, ncand, rng):
    boxes = np.zeros((B, N, 4), np.float32)
    cls = np.zeros((B, N, 1), np.float32)
    for b in range(B):
        idx = rng.choice(N, ncand, replace=False)
        if kind == "disjoint":
            gx = (np.arange(ncand) % 64) * 20.0; gy = (np.arange(ncand) // 64) * 20.0
            bx = np.stack([gx, gy, gx + 16, gy + 16], 1)
        elif kind == "clustered":     # groups of ~6 overlapping boxes
            g = np.arange(ncand) // 6
            gx = (g % 40) * 32.0; gy = (g // 40) * 32.0
            j = rng.uniform(-2, 2, (ncand, 2))
            bx = np.stack([gx + j[:, 0], gy + j[:, 1], gx + 24 + j[:, 0], gy + 24 + j[:, 1]], 1)
        else:                          # heavy: 10 objects, everything piles on them
            g = rng.integers(0, 10, ncand)
            gx = g * 120.0; j = rng.uniform(-6, 6, (ncand, 4))
            bx = np.stack([gx + j[:, 0], 100 + j[:, 1], gx + 100 + j[:, 2], 300 + j[:, 3]], 1)
        boxes[b, idx] = bx
        cls[b, idx, 0] = rng.uniform(0.06, 0.99, ncand)
    return boxes, cls

def main():
    dev = torch.device("cuda", 0)
    h = L.Handle(0)
    h.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    B, N = 8, 200700
    rng = np.random.default_rng(0)
    ws_bytes = L.lib.rtn_detect_workspace_bytes(B, N, 1)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    ob = torch.empty(B, 300, 4, device=dev); os_ = torch.empty(B, 300, device=dev); ol = torch.empty(B, 300, dtype=torch.int32, device=dev)
    for kind in ("disjoint", "clustered", "heavy"):
        for ncand in (300, 1935, 6000):
            bx, cl = make(kind, B, N, ncand, rng)
            tb, tc = torch.from_numpy(bx).to(dev), torch.from_numpy(cl).to(dev)
            def run():
                h.check(L.lib.rtn_filter_detections(h.raw, B, N, 1, tb.data_ptr(), tc.data_ptr(), C.c_float(0.05), C.c_float(0.5), 300,
                                                    ob.data_ptr(), os_.data_ptr(), ol.data_ptr(), ws.data_ptr(), ws_bytes))
            for _ in range(3): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            e0.record()
            for _ in range(reps): run()
            e1.record(); torch.cuda.synchronize()
            kept = int((ol[0] >= 0).sum())
            print("%-10s cand %5d  kept %3d  %.1f us" % (kind, ncand, kept, e0.elapsed_time(e1) / reps * 1e3), flush=True)

main()
