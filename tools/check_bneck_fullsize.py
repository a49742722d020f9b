"""Full-size check of the fused bottleneck path: run-to-run determinism and the difference to the unfused path per stage output."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_fullsize as T
from oracle import ref_numpy as R
pkg = importlib.import_module("retinanet-for-table-detection_amd")
E, Wt = T.mods(pkg)
canvas = (800, 1333)
state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=0.0, tame=True)
x = torch.as_tensor(R.preprocess_custom_tf(T.pages(8, canvas, seed=77).numpy())).cuda()
eng = E.Engine("resnet50", 1, 9, dtype="bf16"); eng.load_state(state)
outs = {}
for fuse in (True, False):
    eng.fuse_bottleneck = fuse
    plan = eng._plan(8, *canvas)
    runs = []
    for rep in range(3):
        for t in plan["feats"]: t.fill_(-7.0)
        reg, cls = eng.forward(x); torch.cuda.synchronize()
        runs.append([t.clone() for t in plan["feats"]] + [reg.clone(), cls.clone()])
    same = all(all(torch.equal(a, b) for a, b in zip(runs[0], r)) for r in runs[1:])
    print("fuse_bottleneck=%s: 3 runs bit-identical: %s" % (fuse, same))
    if not same:
        for i, (a, b) in enumerate(zip(runs[0], runs[1])):
            d = (a.float() - b.float()).abs()
            print("   tensor %d: %d elements differ, max %.4f" % (i, int((d > 0).sum()), float(d.max())))
    outs[fuse] = runs[0]
for i, name in enumerate(["C2", "C3", "C4", "C5", "regression", "classification"]):
    a, b = outs[True][i].float(), outs[False][i].float()
    d = (a - b).abs()
    idx = int(d.argmax())
    print("%s: fused vs unfused max |diff| %.4f (scale %.2f), rms %.5f, differing elements %.3f %%, argmax flat index %d" %
          (name, float(d.max()), float(b.abs().max()), float(torch.sqrt((d ** 2).mean())), 100.0 * float((d > 0).float().mean()), idx))
    if name == "C2":
        # where are the large differences? pixel positions of the worst 5
        top = torch.topk(d.flatten(), 5).indices.cpu().numpy()
        shape = a.shape
        print("   worst C2 elements (b, y, x, c):", [tuple(int(v) for v in np.unravel_index(t, shape)) for t in top])
