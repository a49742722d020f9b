"""In-kernel time line of one K-step group of the head-tower kernel: needs a library built with -DRTN_H8_STAMP (see
csrc/rtn_conv_halo8.hip) passed as RTN_LIB_PATH.  Prints, for wave 0 (leading group) and wave 4 (lagging group) of workgroup 0, the
s_memtime deltas between the four stamps of every phase of three consecutive K steps:
  1 = fragment reads + DMA issued and the reads back (the stamp waits lgkmcnt(0)), 2 = past the first barrier, 3 = MFMAs issued,
  4 = past the second barrier."""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, bench
E = importlib.import_module(bench.PKG + ".engine"); Wt = importlib.import_module(bench.PKG + ".weights"); L = importlib.import_module(bench.PKG + "._lib")
state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=bench.CLS_BIAS, tame=True)
eng = E.Engine("resnet50", 1, 9, dtype="bf16"); eng.load_state(state)
x = bench.synth_images(torch, bench.BATCH, 1000, "cuda")
eng.detect(x); torch.cuda.synchronize()
plan = eng._plan(bench.BATCH, *bench.CANVAS)
op = [o for o in plan["ops"] if o[0] == "conv" and o[2] == "pyramid_regression_1"][0]
for _ in range(3): eng._run_op(op, x)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 128)()
f = L.lib.rtn_debug_h8_stamps
f.argtypes = [C.POINTER(C.c_ulonglong)]; f.restype = C.c_int
assert f(buf) == 0
for w, name in ((0, "wave 0 (leading group)"), (1, "wave 4 (lagging group)")):
    st = [buf[w * 64 + i] for i in range(48)]
    print(name, "- clocks since the group's first stamp; per phase: reads-done / barrier-1 / mfma-issued / barrier-2")
    t0 = st[0]
    for step in range(3):
        row = []
        for ph in range(4):
            a = st[(step * 4 + ph) * 4:(step * 4 + ph) * 4 + 4]
            row.append("%5d %5d %5d %5d" % tuple(v - t0 for v in a))
        print("  tap %d: " % step + " | ".join(row))
    print("  whole group (3 K steps): %d clocks of s_memtime" % (st[47] - st[0]))
