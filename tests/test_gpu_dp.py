"""Device-side data-parallel equivalence on ONE GPU: two ranks (two processes sharing device 0, gloo in place of RCCL, which
refuses two ranks on one device) each run Trainer.forward_backward + optimizer_step on HALF of a batch; the reduced flat
gradient and the updated master weights must equal a single-rank run on the MERGED batch.

This is what keras.utils.multi_gpu_model defines (RetinaNet.py:105-116): identical weights on every replica, the minibatch sliced
on axis 0, ONE loss over the concatenated outputs - so focal and smooth-L1 are normalised by the positive-anchor count of the
WHOLE batch (model/losses.py:39-44,87-90).  One image of the batch has no ground-truth box (zero positives): a per-rank
normaliser would be wrong for its rank, and the test checks that it would.

fp32 path; stated tolerance: gradients within 1e-5 of the largest gradient element per layer group (the two halves are summed
in another order than the merged batch's pixel splits: fp32 association, not run-to-run noise - the weight gradients have had
no float atomics since round 2), weights after the step within 1e-3 x lr of each other, and bit-identical between the two ranks."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
PKG = "retinanet-for-table-detection_amd"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CANVAS = (128, 160)
B_TOTAL = 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch():
    """Deterministic batch of 4 pages + ground truth; image 1 has NO box (its anchors are all background / ignore)."""
    g = torch.Generator().manual_seed(123)
    x = (torch.rand(B_TOTAL, CANVAS[0], CANVAS[1], 3, generator=g) * 2 - 1)
    rng = np.random.RandomState(7)
    gb = np.zeros((B_TOTAL, 64, 4))
    gc = np.zeros(B_TOTAL, np.int32)
    for b in range(B_TOTAL):
        n = 0 if b == 1 else int(rng.randint(1, 4))
        w, h = rng.uniform(30, 110, n), rng.uniform(25, 90, n)
        x1, y1 = rng.uniform(0, CANVAS[1] - w), rng.uniform(0, CANVAS[0] - h)
        gb[b, :n] = np.stack([x1, y1, x1 + w, y1 + h], axis=1) if n else 0
        gc[b] = n
    return x, gb, gc


def _step(E, T, L, Wt, x, gb, gc, pg, device):
    """One training step on (x, gb, gc); returns (flat gradient after the reduction, master weights after the step, loss sums)."""
    import ctypes as C
    state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=-2.0, tame=True)
    eng = E.Engine("resnet50", 1, 9, dtype="f32", device=0)
    eng.load_state(state)
    tr = T.Trainer(eng, lr=1e-4, clipnorm=0.001, process_group=pg)
    B = x.shape[0]
    cfg, N = E.make_anchor_cfg(CANVAS)
    xd = x.to(device)
    gbd = torch.as_tensor(gb).to(device)
    gld = torch.zeros(B, 64, dtype=torch.int32, device=device)
    gcd = torch.as_tensor(gc).to(device)
    hw = torch.as_tensor(np.tile(np.array(CANVAS, np.int32), (B, 1))).to(device)
    reg_t = torch.empty(B, N, 5, device=device)
    lab_t = torch.empty(B, N, 2, device=device)
    eng._bind_stream()
    eng.h.check(L.lib.rtn_anchor_targets(eng.h.raw, C.byref(cfg), B, 1, gbd.data_ptr(), gld.data_ptr(), gcd.data_ptr(), hw.data_ptr(),
                                         0.4, 0.5, reg_t.data_ptr(), lab_t.data_ptr()))
    tr.forward_backward(xd, reg_t, lab_t)
    own = tr.loss_sums.cpu().numpy().copy()
    tr.optimizer_step()
    torch.cuda.synchronize()
    return tr.grad.cpu().numpy().copy(), tr.master.cpu().numpy().copy(), own, tr.norm_sums.cpu().numpy().copy(), tr


def _worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)                                   # both ranks share the one GPU of the box
    device = torch.device("cuda", 0)
    E, T, L, Wt = [importlib.import_module(PKG + "." + m) for m in ("engine", "trainer", "_lib", "weights")]
    x, gb, gc = _batch()
    per = B_TOTAL // world
    sl = slice(rank * per, (rank + 1) * per)
    g, w, own, merged, tr = _step(E, T, L, Wt, x[sl], gb[sl], gc[sl], dist.group.WORLD, device)
    assert tr.bucketer is not None and len(tr.bucketer.buckets) >= 2
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), grad=g, master=w, own=own, merged=merged)
    dist.barrier()
    if rank == 0:                                              # the merged batch on one rank, no process group
        g1, w1, own1, merged1, tr1 = _step(E, T, L, Wt, x, gb, gc, None, device)
        layout = {k: (v["woff"], v["woff"] + v["rows"] * v["K"]) for k, v in tr1.eng.layout.items()}
        np.savez(os.path.join(outdir, "single.npz"), grad=g1, master=w1, own=own1, names=np.array(list(layout)),
                 spans=np.array(list(layout.values())), NW=tr1.NW)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_halves_equal_one_rank_on_the_merged_batch(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1, one = [np.load(str(tmp_path / n)) for n in ("rank0.npz", "rank1.npz", "single.npz")]
    # merged-batch normaliser: the all-reduced sums are the single-rank sums; rank 0 holds the image without positives
    assert np.allclose(r0["merged"], one["own"], rtol=1e-6) and np.array_equal(r0["merged"], r1["merged"])
    assert r0["own"][2] < one["own"][2] and r1["own"][2] < one["own"][2] and r0["own"][2] + r1["own"][2] == one["own"][2]
    assert one["own"][2] >= 8                                   # enough positives for a meaningful gradient
    # identical replicas: same reduced gradient, same weights after the step, bit for bit
    assert np.array_equal(r0["grad"], r1["grad"]) and np.array_equal(r0["master"], r1["master"])
    # reduced gradient == gradient of the merged batch, layer by layer
    worst = (0.0, "")
    for name, (lo, hi) in zip(one["names"], one["spans"]):
        a, b = r0["grad"][lo:hi], one["grad"][lo:hi]
        scale = float(np.abs(b).max())
        if scale == 0.0:
            assert not a.any()
            continue
        err = float(np.abs(a - b).max()) / scale
        worst = max(worst, (err, str(name)))
        assert err <= 1e-5, "%s: reduced gradient differs from the merged-batch gradient by %.3e of its scale" % (name, err)
    nb = int(one["NW"])
    bscale = float(np.abs(one["grad"][nb:]).max())
    assert bscale > 0 and float(np.abs(r0["grad"][nb:] - one["grad"][nb:]).max()) <= 1e-5 * bscale
    print("worst layer %s: %.3e of the gradient scale" % (worst[1], worst[0]))
    # a per-rank normaliser would have scaled rank 0's gradient by merged/own positives: visibly different
    assert one["own"][2] / max(1.0, r0["own"][2]) > 1.2
    # weights after the clipped Adam step
    assert float(np.abs(r0["master"] - one["master"]).max()) <= 1e-3 * 1e-4
