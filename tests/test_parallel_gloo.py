"""CPU tests (gloo, world_size 2) of the data-parallel layer: bucketed gradient all-reduce and the merged-batch loss
normaliser that reproduces keras.utils.multi_gpu_model's single loss over the concatenated batch (RetinaNet.py:105-116)."""
import importlib
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_numpy as R

PKG = "retinanet-for-table-detection_amd"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    Par = importlib.import_module(PKG + ".parallel")
    # ---- 1. bucketer: segments finish in reverse (backward) order, a few out of order
    n = 10000
    segs, o = [], 0
    rng = np.random.RandomState(0)
    for i in range(23):
        ln = int(rng.randint(50, 900))
        segs.append(("L%d" % i, o, o + ln))
        o += ln
    flat = torch.arange(o, dtype=torch.float32) * (rank + 1)
    b = Par.GradBucketer(flat, segs, bucket_bytes=4000)
    assert len(b.buckets) > 3
    order = [s[0] for s in reversed(segs)]
    order[3], order[7] = order[7], order[3]
    for name in order[:-2]:
        b.layer_done(name)
    b.finish()                                                   # launches the unfinished buckets too
    want = torch.arange(o, dtype=torch.float32) * sum(r + 1 for r in range(world))
    assert torch.equal(flat, want)
    # second use after reset
    flat.fill_(rank + 1.0)
    for name in order:
        b.layer_done(name)
    b.finish()
    assert torch.all(flat == sum(r + 1 for r in range(world)))
    # ---- 2. merged-batch normaliser: per-rank loss gradients with all-reduced counts == gradients of the whole batch
    rng = np.random.RandomState(42)
    B, N, K = 4, 600, 1
    state = rng.choice([-1.0, 0.0, 1.0], size=(B, N), p=[0.1, 0.8, 0.1])
    state[0] = np.where(state[0] == 1, 0, state[0])              # image 0 has NO positives: its rank's own count differs
    lab = np.zeros((B, N, 2)); lab[..., 1] = state; lab[..., 0] = state == 1
    p = rng.uniform(0.01, 0.99, size=(B, N, K))
    regt = np.zeros((B, N, 5)); regt[..., :4] = rng.normal(size=(B, N, 4)); regt[..., 4] = state
    pred = regt[..., :4] + rng.normal(scale=0.2, size=(B, N, 4))
    sl = slice(rank * B // world, (rank + 1) * B // world)
    fs, npos, gcls = R.focal_loss(lab[sl], p[sl], grad=True)
    rs, nposr, greg = R.smooth_l1_loss(regt[sl], pred[sl], grad=True)
    sums = Par.allreduce_loss_sums(torch.tensor([fs, rs, npos, nposr], dtype=torch.float64))
    Fs, NP, Gcls = R.focal_loss(lab, p, grad=True)
    Rs, NPR, Greg = R.smooth_l1_loss(regt, pred, grad=True)
    assert abs(sums[0].item() - Fs) < 1e-9 * abs(Fs) and sums[2].item() == NP and sums[3].item() == NPR
    mine_cls = gcls / max(1.0, sums[2].item())
    mine_reg = greg / max(1.0, sums[3].item())
    assert np.allclose(mine_cls, Gcls[sl] / max(1, NP), rtol=0, atol=1e-15)
    assert np.allclose(mine_reg, Greg[sl] / max(1, NPR), rtol=0, atol=1e-15)
    own = gcls / max(1.0, npos)                                  # the WRONG (per-rank) normaliser really differs
    assert not np.allclose(own, mine_cls)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(outdir, "ok%d" % rank), "w").write("ok")


def test_data_parallel_layer_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def test_bucketer_waits_for_every_lane_of_a_bucket(monkeypatch):
    """GradBucketer's event logic without a GPU: a recording stub stands in for torch.cuda (streams with a logical clock, events
    that remember when they were recorded) and for dist.all_reduce.  The trainer spreads weight gradients over three lanes; a
    bucket's all-reduce must be issued on the communication stream behind one event per lane that produced one of its layers, each
    recorded AFTER that lane's last layer of the bucket, and the bias segment (fed from every lane) behind all of them."""
    import importlib
    import contextlib
    import torch
    P = importlib.import_module("retinanet-for-table-detection_amd.parallel")

    class Stream:
        def __init__(self, device=None, sid=None):
            self.cuda_stream = sid if sid is not None else 900 + len(layer.streams)
            self.clock, self.waits = 0, []
            layer.streams.append(self)

        def wait_event(self, ev):
            self.waits.append((ev.stream.cuda_stream, ev.at))

        def wait_stream(self, other):
            self.waits.append(("stream", other.cuda_stream))

    class Event:
        def __init__(self):
            self.stream, self.at = None, None

        def record(self, st):
            self.stream, self.at = st, st.clock

    class Layer:
        streams = []

        def __init__(self):
            self.cur = None

        def current_stream(self, device=None):
            return self.cur

        @contextlib.contextmanager
        def stream(self, st):
            prev, self.cur = self.cur, st
            try:
                yield
            finally:
                self.cur = prev
    Layer.Stream, Layer.Event = Stream, Event
    layer = Layer()
    lanes = {i: Stream(sid=i) for i in (0, 1, 2, 3)}
    layer.cur = lanes[0]
    calls = []

    class Work:
        def wait(self):
            pass

    def fake_all_reduce(view, op=None, group=None, async_op=False):
        calls.append((view.numel(), list(layer.cur.waits)))
        return Work()
    monkeypatch.setattr(P.dist, "all_reduce", fake_all_reduce)
    flat = torch.zeros(100)
    segs = [("a", 0, 20), ("b", 20, 40), ("c", 40, 60), ("d", 60, 80), ("__biases__", 80, 100)]
    bk = P.GradBucketer(flat, segs, group=None, bucket_bytes=160, stream_layer=layer)        # 40 floats per bucket
    assert [sorted(s_[0] for s_ in b) for b in bk.buckets] == [["__biases__", "d"], ["b", "c"], ["a"]]

    def produce(name, lane):            # a layer's gradient kernels enqueued on `lane`, then reported there
        lanes[lane].clock += 1
        with layer.stream(lanes[lane]):
            bk.layer_done(name)
    # backward order: d (lane 1), c (lane 2), b (lane 3), then the biases' last fused gradient on lane 1 after every lane was marked
    produce("d", 1)
    assert not calls                                        # the bias segment is still pending
    produce("c", 2)
    produce("b", 3)
    assert len(calls) == 1 and calls[0][0] == 40            # bucket {b, c}: behind lane 2 at its clock 1 and lane 3 at its clock 1
    assert sorted(calls[0][1]) == [(2, 1), (3, 1)]
    for ln in (1, 2, 3):                                    # the fused bias gradients ran on every lane: mark them all, then report
        lanes[ln].clock += 1
        with layer.stream(lanes[ln]):
            bk.mark("__biases__")
    with layer.stream(lanes[1]):
        bk.layer_done("__biases__")
    assert len(calls) == 2 and calls[1][0] == 40
    new_waits = calls[1][1][len(calls[0][1]):]              # the communication stream's waits accumulate: the second bucket's are the tail
    assert sorted(new_waits) == [(1, 2), (2, 2), (3, 2)]    # every lane, each at its LATEST clock (lane 1's event was re-recorded after d)
    lanes[0].clock += 1
    with layer.stream(lanes[0]):
        bk.finish()                                         # bucket {a} was never reported: goes out behind the caller's stream
    assert len(calls) == 3 and calls[2][0] == 20 and (0, 1) in calls[2][1]
    assert ("stream", bk.comm.cuda_stream) in lanes[0].waits
    assert bk.last_marks[0] == {1, 2, 3} and bk.last_marks[1] == {2, 3}
