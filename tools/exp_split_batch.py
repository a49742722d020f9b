"""Experiment: does running the batch as S independent sub-batches on S HIP streams beat one launch chain?
(tails of each layer's last partly-filled round of workgroups are filled by the other sub-batch's kernels)"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as Bn
pkg = importlib.import_module("retinanet-for-table-detection_amd")
E = importlib.import_module("retinanet-for-table-detection_amd.engine")
Wt = importlib.import_module("retinanet-for-table-detection_amd.weights")

def run(S, B=8, steps=20, warmup=3):
    dev = torch.device("cuda", 0)
    state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=Bn.CLS_BIAS, tame=True)
    engs = [E.Engine("resnet50", 1, 9, dtype="bf16", device=0) for _ in range(S)]
    for e in engs:
        e.load_state(state)
    x = Bn.synth_images(torch, B, 1, dev)
    parts = list(x.chunk(S))
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    main = torch.cuda.current_stream(dev)
    fork, joins = torch.cuda.Event(), [torch.cuda.Event() for _ in range(S)]
    def step():
        if S == 1:
            engs[0].detect(parts[0]); return
        fork.record(main)
        for s in range(S):
            streams[s].wait_event(fork)
            with torch.cuda.stream(streams[s]):
                engs[s].detect(parts[s])
                joins[s].record(streams[s])
        for s in range(S):
            main.wait_event(joins[s])
    for _ in range(warmup): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print("sub-batches %d: %.3f ms/step  %.1f img/s" % (S, ms, B / ms * 1e3), flush=True)

for S in (1, 2, 4, 1, 2):
    run(S)
