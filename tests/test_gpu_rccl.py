"""The RCCL code path on the one GPU of the test box: init_process_group("nccl", device_id=...) with world_size 1 — the exact call
bench.py makes under torch.distributed.run — then one Trainer step with the gradient bucketer on (bucketed async all-reduce on the
communication stream behind per-lane events, all-reduced loss sums, barrier).  With one rank every collective is the identity, so
the step must equal the step of a Trainer without a process group BIT FOR BIT (the weight gradients have no float atomics), in
bf16 with the three weight-gradient lanes running.  This is what keras.utils.multi_gpu_model's role becomes here (RetinaNet.py:
105-116); the 2-rank equivalence itself is tests/test_gpu_dp.py (gloo: RCCL refuses two ranks on one device) and
tests/test_parallel_gloo.py (CPU)."""
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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, port, outdir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = "0", "1", "0"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import ctypes as C
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))          # bench.py's call
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    device = torch.device("cuda", 0)
    E, T, L, Wt = [importlib.import_module(PKG + "." + m) for m in ("engine", "trainer", "_lib", "weights")]
    state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=-2.0, tame=True)
    g = torch.Generator().manual_seed(5)
    B = 2
    x = (torch.rand(B, CANVAS[0], CANVAS[1], 3, generator=g) * 2 - 1).to(device)
    cfg, N = E.make_anchor_cfg(CANVAS)
    gb = np.zeros((B, 64, 4))
    gb[0, 0], gb[1, 0], gb[1, 1] = [10, 12, 90, 70], [40, 30, 150, 120], [5, 60, 60, 120]
    gbd, gld = torch.as_tensor(gb).to(device), torch.zeros(B, 64, dtype=torch.int32, device=device)
    gcd = torch.as_tensor(np.array([1, 2], np.int32)).to(device)
    hw = torch.as_tensor(np.tile(np.array(CANVAS, np.int32), (B, 1))).to(device)
    reg_t, lab_t = torch.empty(B, N, 5, device=device), torch.empty(B, N, 2, device=device)
    out = {}
    for tag, pg in (("rccl", dist.group.WORLD), ("none", None)):
        eng = E.Engine("resnet50", 1, 9, dtype="bf16", device=0)
        eng.load_state(state)
        tr = T.Trainer(eng, lr=1e-4, clipnorm=0.001, process_group=pg)
        eng._bind_stream()
        eng.h.check(L.lib.rtn_anchor_targets(eng.h.raw, C.byref(cfg), B, 1, gbd.data_ptr(), gld.data_ptr(), gcd.data_ptr(), hw.data_ptr(),
                                             0.4, 0.5, reg_t.data_ptr(), lab_t.data_ptr()))
        losses = []
        for _ in range(2):                                     # two steps: the bucketer is reset and re-used
            losses.append(tr.train_on_batch(x, reg_t, lab_t))
        torch.cuda.synchronize()
        if pg is not None:
            assert tr.bucketer is not None and len(tr.bucketer.buckets) >= 2
            assert tr.wgrad_lanes == 3                         # the lanes stay on under data parallelism
            # With one rank the all-reduce is the identity, so a missing event wait could not change a bit: check the waits themselves.
            # Every bucket's all-reduce must have been issued behind an event on EVERY stream that produced one of its layers'
            # weight gradients (GradBucketer.last_marks = the streams each bucket waited for in the last step), and the bias
            # bucket - fed by fused bias gradients on every weight-gradient lane - behind all of those lanes.
            bp = tr._bplan(B, CANVAS[0], CANVAS[1])
            sch = bp[("bsched", tr.wgrad_lanes)]
            main = torch.cuda.current_stream(device)
            streams = [main] + tr._wg_stream[:sch["nlanes"] - 1]
            need = [set() for _ in tr.bucketer.buckets]
            wg_lanes = set()
            for bi_, b_ in enumerate(bp["bops"]):
                if b_[0] == "wgrad":
                    sid = streams[sch["lanes"][bi_]].cuda_stream
                    wg_lanes.add(sid)
                    if b_[3] in tr.bucketer.where:
                        need[tr.bucketer.where[b_[3]]].add(sid)
            need[tr.bucketer.where["__biases__"]] |= {sid for sid in wg_lanes if sid != main.cuda_stream}
            assert len(wg_lanes) >= 3
            for bi_, want_ in enumerate(need):
                assert want_ <= tr.bucketer.last_marks[bi_], "bucket %d went out without waiting for streams %s" % (bi_, want_ - tr.bucketer.last_marks[bi_])
            t = torch.ones(4, device=device)
            dist.all_reduce(t)
            dist.barrier()
            assert float(t.sum()) == 4.0
        out[tag + "_grad"] = tr.grad.cpu().numpy()
        out[tag + "_master"] = tr.master.cpu().numpy()
        out[tag + "_loss"] = np.array(losses)
    np.savez(os.path.join(outdir, "rccl1.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


def test_one_rank_rccl_step_equals_the_step_without_a_process_group(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)          # a fresh process: RCCL initialises its own context
    r = np.load(str(tmp_path / "rccl1.npz"))
    assert np.isfinite(r["rccl_loss"]).all() and np.array_equal(r["rccl_loss"], r["none_loss"])
    assert float(np.abs(r["none_grad"]).max()) > 0
    assert np.array_equal(r["rccl_grad"], r["none_grad"]), "max difference %.3e" % float(np.abs(r["rccl_grad"] - r["none_grad"]).max())
    assert np.array_equal(r["rccl_master"], r["none_master"])
