"""Per-op device time of one training step (forward + loss + backward + optimizer) at 800x1333."""
import importlib, os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np, torch, bench
PKG = bench.PKG
E = importlib.import_module(PKG + ".engine"); Wt = importlib.import_module(PKG + ".weights"); T = importlib.import_module(PKG + ".trainer"); L = importlib.import_module(PKG + "._lib")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=-2.0, tame=True)
eng = E.Engine("resnet50", 1, 9, dtype=dtype); eng.load_state(state)
tr = T.Trainer(eng)
x = bench.synth_images(torch, B, 1000, "cuda")
if dtype == "f32": x = x.float()
# targets on the device through rtn_anchor_targets
cfg, N = E.make_anchor_cfg(bench.CANVAS)
rng = np.random.RandomState(0)
gb = np.zeros((B, 64, 4)); gc = np.zeros(B, np.int32)
for b in range(B):
    g = rng.randint(1, 7); w, h = rng.uniform(80, 900, g), rng.uniform(60, 600, g)
    x1, y1 = rng.uniform(0, 1333 - w), rng.uniform(0, 800 - h)
    gb[b, :g] = np.stack([x1, y1, x1 + w, y1 + h], 1); gc[b] = g
gbd, gld, gcd = torch.as_tensor(gb).cuda(), torch.zeros(B, 64, dtype=torch.int32, device="cuda"), torch.as_tensor(gc).cuda()
hw = torch.as_tensor(np.tile(np.array(bench.CANVAS, np.int32), (B, 1))).cuda()
reg_t = torch.empty(B, N, 5, device="cuda"); lab_t = torch.empty(B, N, 2, device="cuda")
eng._bind_stream()
eng.h.check(L.lib.rtn_anchor_targets(eng.h.raw, C.byref(cfg), B, 1, gbd.data_ptr(), gld.data_ptr(), gcd.data_ptr(), hw.data_ptr(), 0.4, 0.5, reg_t.data_ptr(), lab_t.data_ptr()))
for _ in range(2): tr.train_on_batch(x, reg_t, lab_t)
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 5
for _ in range(n): tr.forward_backward(x, reg_t, lab_t); tr.optimizer_step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print("train step: %.2f ms for batch %d -> %.1f images/s" % (dt * 1e3, B, B / dt))
# phase timing with events
def timed(fn):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); fn(); e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)
print("forward      %.3f ms" % timed(lambda: eng.forward(x)))
bp = tr._bplan(B, *bench.CANVAS)
acc = collections.OrderedDict()
tr.grad.zero_()
tabs = {}
for b in bp["bops"]:
    if b[0] == "wgrad":
        tab = torch.empty(L.lib.rtn_conv2d_wgrad_workspace_bytes(C.byref(b[1])), dtype=torch.uint8, device="cuda")
        eng.h.check(L.lib.rtn_conv2d_wgrad_rowinfo(eng.h.raw, C.byref(b[1]), tab.data_ptr(), tab.numel()))
        tabs[id(b)] = tab
torch.cuda.synchronize()
for b in bp["bops"]:
    kind = b[0]
    h, lib = eng.h, L.lib
    def run():
        if kind == "wgrad":          # as the trainer runs it: on the row-info table built once per layer
            h.check(lib.rtn_conv2d_wgrad_prepared(h.raw, C.byref(b[1]), b[2].data_ptr(), b[4].data_ptr() if b[4] is not None else None,
                                                  b[5] if b[4] is not None else 0, tabs[id(b)].data_ptr(), tabs[id(b)].numel()))
        elif kind == "dgrad": h.check(lib.rtn_conv2d_dgrad(h.raw, C.byref(b[1])))
        elif kind == "bgrad": h.check(lib.rtn_bias_grad(h.raw, b[1].data_ptr(), eng.rdt, b[2], b[3], b[4], b[5].data_ptr()))
        elif kind == "padcast": h.check(lib.rtn_pad_cast_rows(h.raw, b[1].data_ptr(), b[2].data_ptr(), eng.rdt, b[3], b[4], b[5]))
        elif kind == "zins": h.check(lib.rtn_zero_insert2(h.raw, b[1].data_ptr(), b[2].data_ptr(), eng.rdt, *b[3]))
        elif kind == "upbwd": h.check(lib.rtn_upsample_add_bwd(h.raw, b[1].data_ptr(), b[2].data_ptr(), eng.rdt, *b[3], b[4]))
        elif kind == "poolbwd":
            fused = eng.fuse_stem and eng.fuse_stem_train and eng.dtype == "bf16"
            h.check(lib.rtn_maxpool3x3s2_tfsame_bwd_idx(h.raw, b[2].data_ptr(), b[5].data_ptr(), (b[6] if fused else b[1]).data_ptr(), b[3].data_ptr(), eng.rdt, *b[4], 2 if fused else 1))
    ms = timed(run)
    label = kind + ((":" + (b[3] if kind == "wgrad" else b[-1])) if kind in ("wgrad", "dgrad") else "")
    acc[label] = acc.get(label, 0) + ms
tot = collections.Counter()
for k, v in acc.items(): tot[k.split(":")[0]] += v
print("backward by kind:", {k: round(v, 3) for k, v in tot.items()}, "sum %.3f ms" % sum(tot.values()))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:400]: print("  %-40s %.3f ms" % (k, v))
print("optimizer    %.3f ms" % timed(lambda: tr.optimizer_step()))
