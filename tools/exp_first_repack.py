"""Why did one dispatch of pack_dgrad_multi_kernel take 60 ms in the round-3 training trace (profiles/r3_v3_train_kernel_stats.csv: 8 calls,
min 84 us, max 60.2 ms)?  The first call of a Trainer, timed here with events in fresh states."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, bench
E = importlib.import_module(bench.PKG + ".engine"); Wt = importlib.import_module(bench.PKG + ".weights"); T = importlib.import_module(bench.PKG + ".trainer")
state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=-2.0, tame=True)
def ev_ms(fn):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); fn(); e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)
mode = sys.argv[1] if len(sys.argv) > 1 else "cold"
eng = E.Engine("resnet50", 1, 9, dtype="bf16"); eng.load_state(state)
if mode == "warm":                       # the GPU busy for a while first: clocks up, code object resident
    x = torch.randn(8192, 8192, device="cuda")
    for _ in range(20): y = x @ x
    torch.cuda.synchronize()
orig = T.Trainer._repack_dgrad
times = []
def timed_repack(self):
    times.append(ev_ms(lambda: orig(self)))
T.Trainer._repack_dgrad = timed_repack
tr = T.Trainer(eng, lr=1e-4, clipnorm=0.001)
tr._repack_dgrad(); tr._repack_dgrad()
time.sleep(3.0)                          # idle: does the next call pay again?
tr._repack_dgrad(); tr._repack_dgrad()
print("%s process: repack #1 (inside Trainer.__init__, fresh dgrad-weight buffers) %.3f ms, #2 %.3f, #3 %.3f, after 3 s idle %.3f, then %.3f" % ((mode,) + tuple(times)))
