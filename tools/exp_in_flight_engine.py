"""Engine.in_flight against the two-engine experiment, same process."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, bench
E = importlib.import_module(bench.PKG + ".engine"); Wt = importlib.import_module(bench.PKG + ".weights")
state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=bench.CLS_BIAS, tame=True)
x = bench.synth_images(torch, bench.BATCH, 1, "cuda")
def timed(fn, n=60):
    for _ in range(6): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
eng = E.Engine("resnet50", 1, 9, dtype="bf16"); eng.load_state(state)
print("one at a time, lanes on, default stream: %.3f ms" % timed(lambda: eng.detect(x)), flush=True)
for n in (2, 3):
    eng.in_flight = n
    print("Engine.in_flight=%d, caller = default stream: %.3f ms" % (n, timed(lambda: eng.detect(x))), flush=True)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        print("Engine.in_flight=%d, caller = a side stream:  %.3f ms" % (n, timed(lambda: eng.detect(x))), flush=True)
for n in (2, 3):
    eng.in_flight = n
    for rep in range(2):
        for traced in (True, False):
            for _ in range(6): eng.detect(x)
            torch.cuda.synchronize()
            eng.trace_in_flight = [] if traced else None
            ref = torch.cuda.Event(enable_timing=True); ref.record()
            t0 = time.perf_counter()
            for _ in range(60): eng.detect(x)
            torch.cuda.synchronize()
            host = (time.perf_counter() - t0) / 60 * 1e3
            if traced:
                tl = [(si, ref.elapsed_time(a), ref.elapsed_time(b)) for si, a, b in eng.trace_in_flight]
                print("in_flight=%d traced:   %.3f ms/step by the host clock; batch durations %s ... %s" % (
                    n, host, " ".join("%.2f" % (b - a) for _, a, b in tl[:6]), " ".join("%.2f" % (b - a) for _, a, b in tl[-4:])), flush=True)
            else:
                print("in_flight=%d untraced: %.3f ms/step by the host clock" % (n, host), flush=True)
            eng.trace_in_flight = None
eng.in_flight = 1
engs = [E.Engine("resnet50", 1, 9, dtype="bf16") for _ in range(2)]
for e in engs: e.load_state(state); e.two_streams = False
streams = [torch.cuda.Stream() for _ in range(2)]
cnt = [0]
def two():
    i = cnt[0] & 1; cnt[0] += 1
    with torch.cuda.stream(streams[i]): engs[i].detect(x)
print("two engines on two streams, lanes off: %.3f ms" % timed(two), flush=True)
