"""Experiment: N independent batch-8 steps in flight (N engines = N buffer sets, one HIP stream each unless RTN_TWO_STREAMS=1)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, bench
E = importlib.import_module(bench.PKG + ".engine"); Wt = importlib.import_module(bench.PKG + ".weights")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=bench.CLS_BIAS, tame=True)
dev = torch.device("cuda", 0)
engs = [E.Engine("resnet50", 1, 9, dtype="bf16") for _ in range(N)]
for e in engs: e.load_state(state)
xs = [bench.synth_images(torch, bench.BATCH, 1 + i, "cuda") for i in range(N)]
streams = [torch.cuda.Stream(device=dev) for _ in range(N)]
def run(n):
    for i in range(2 * N):
        with torch.cuda.stream(streams[i % N]): engs[i % N].detect(xs[i % N])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        with torch.cuda.stream(streams[i % N]): engs[i % N].detect(xs[i % N])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(2):
    t = run(60)
    print("%d in flight (lanes %s): %.3f ms/step = %.0f img/s" % (N, os.environ.get("RTN_TWO_STREAMS", "1"), t, bench.BATCH / t * 1e3), flush=True)
