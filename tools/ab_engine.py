"""Same-process A/B of whole inference steps: one engine, its knobs flipped between interleaved rounds (run-to-run noise between
bench.py processes is +-1.5 %, more than most kernel changes).  A variant is ATTR=VAL / ENV=VAL pairs joined by '+': lower-case names
are Engine attributes (fuse_chain=0), upper-case names environment knobs of librtn.so (RTN_CHAIN_SPREAD=0).
   python tools/ab_engine.py [--in-flight F] [--rounds R] [--steps S] VARIANT VARIANT ...     ('-' = the defaults)"""
import argparse, importlib, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, bench
ap = argparse.ArgumentParser(); ap.add_argument("--in-flight", type=int, default=2); ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--steps", type=int, default=40); ap.add_argument("variants", nargs="+")
args = ap.parse_args()
E = importlib.import_module(bench.PKG + ".engine"); Wt = importlib.import_module(bench.PKG + ".weights")
state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=bench.CLS_BIAS, tame=True)
x = bench.synth_images(torch, bench.BATCH, 1, "cuda")
eng = E.Engine("resnet50", 1, 9, dtype="bf16"); eng.load_state(state); eng.in_flight = args.in_flight
variants = [dict(kv.split("=") for kv in v.split("+")) if v != "-" else {} for v in args.variants]
attrs = sorted({k for v in variants for k in v if k.islower()}); envs = sorted({k for v in variants for k in v if not k.islower()})
defaults = {k: getattr(eng, k) for k in attrs}
def apply(v):
    eng.join()
    for k in attrs: setattr(eng, k, type(defaults[k])(int(v[k])) if k in v else defaults[k])
    for k in envs:
        os.environ.pop(k, None)
        if k in v: os.environ[k] = v[k]
times = [[] for _ in variants]
for rnd in range(args.rounds + 1):
    for i, v in enumerate(variants):
        apply(v)
        for _ in range(6): eng.detect(x)
        eng.join(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(args.steps): eng.detect(x)
        eng.join(); torch.cuda.synchronize()
        if rnd: times[i].append((time.perf_counter() - t0) / args.steps * 1e3)
print("batch %d, %d in flight, %d rounds x %d steps, ms per step (median, min-max) and img/s:" % (bench.BATCH, args.in_flight, args.rounds, args.steps))
for name, ts in zip(args.variants, times):
    m = statistics.median(ts)
    print("  %-44s %.3f  (%.3f-%.3f)  %.0f img/s" % (name, m, min(ts), max(ts), bench.BATCH / m * 1e3))
