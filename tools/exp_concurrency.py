"""Do two launches on two HIP streams overlap?  One conv layer of the bench plan, N launches on one stream vs N/2 on each of two
streams (two engines: separate buffers), with the persistent kernels' grid limited (RTN_CONV_H8_GRID) so that half the chip is free."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, bench
E = importlib.import_module(bench.PKG + ".engine"); Wt = importlib.import_module(bench.PKG + ".weights")
name = sys.argv[1] if len(sys.argv) > 1 else "pyramid_regression_1"
state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=bench.CLS_BIAS, tame=True)
engs = [E.Engine("resnet50", 1, 9, dtype="bf16") for _ in range(2)]
xs = [bench.synth_images(torch, bench.BATCH, 1 + i, "cuda") for i in range(2)]
ops = []
for e, x in zip(engs, xs):
    e.load_state(state); e.detect(x); torch.cuda.synchronize()
    ops.append({op[2]: op for op in e._plan(bench.BATCH, *bench.CANVAS)["ops"] if op[0] == "conv"}[name])
streams = [torch.cuda.Stream() for _ in range(2)]
def run(two, n=40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        k = (i & 1) if two else 0
        with torch.cuda.stream(streams[k]):
            engs[k]._bind_stream(); engs[k]._run_op(ops[k], xs[k])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for g in (0, 128, 64):
    os.environ["RTN_CONV_H8_GRID"] = str(g)
    run(False, 10); run(True, 10)
    print("%s grid limit %3d: one stream %.4f ms/launch, alternating two streams %.4f ms/launch" % (name, g, run(False), run(True)))
