"""bench.py — images/sec of the RetinaNet R50-FPN detection path at 800x1333 on MI355X.

  python bench.py --gpus N --steps K --warmup W
(N > 1: one rank per GPU.  Under torch.distributed.run (the driver's launch) each rank runs main(); run alone with --gpus N > 1 it
starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child process itself.  WORLD_SIZE != --gpus: exit 2.)

Workload (BASELINE.json configs[1]): ResNet50-FPN RetinaNet INFERENCE, 800x1333, bf16, batch 8 per GPU:
one step = stem pack -> ResNet-50 -> FPN -> both head stacks over P3..P7 -> decode + score threshold + NMS + top-k
for a batch of 8 synthetic 3-channel distance-transform-like pages already resident in HBM.  --in-flight N (default 2): the
engine's throughput mode - consecutive steps go to N buffer sets on N HIP streams and overlap on the device (every step still
runs whole, into its own outputs; the timed region ends when all K steps are done); --in-flight 1 is one batch at a time, whose
ms per step the line also carries (`config.ms_per_step_one_batch_at_a_time`).  Inference shards by
image with no data-path collective ("scaling": "weak": per-GPU batch fixed).  Weights are seeded random-init of the
architecture (no checkpoint exists in the reference); the classification bias is set so ~1 % of the 200,700 anchors
per image pass the 0.05 score threshold, the load a trained detector puts on the NMS stage.

The JSON line also carries
  roofline     — bound "mfma": `achieved` = algorithmic conv FLOPs of a step / the TIMED region's time per step (the number the
                 driver's clock anchors; stream lanes overlap launches there), `frac` against the 2.5 PF dense bf16 peak;
                 `dominant` = the kernel with the largest share of GPU time (the head-tower layers), with its own FLOPs per
                 launch and its launch duration measured live with events on its launch stream, launches one after another
                 (`serial_*` fields: the same serial measurement summed over all conv launches — it exceeds ms_per_step
                 because the timed region overlaps lanes); `traffic` = HBM bytes per step from the committed PMC passes, only
                 when that file was collected on the same launch plan (else null, with the reason).
  cpu_baseline — the oracle (oracle/ref_net.py + ref_numpy.py: torch-CPU fp32 restatement, kind "port") timed on this
                 host's cores on a bounded sample (whole path for 1 image of the same canvas).
  secondary    — BASELINE.json configs[2]/[3]: the batch-16-per-GPU TRAINING step (targets + forward + focal/smooth-L1 +
                 backward + bucketed gradient all-reduce under DP + clipnorm Adam), same timing protocol, 20 steps.
"""
import argparse
import importlib
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL / cross-process GPU sharing on this pool needs dmabuf IPC
PKG = "retinanet-for-table-detection_amd"

CANVAS = (800, 1333)
BATCH = 8
GFLOP_PER_IMAGE = 416.1            # 208.06 GMAC forward, every conv counted
BF16_DENSE_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
CAND_FRACTION = 0.01               # share of anchors above the 0.05 score threshold (set by calibration)
CLS_BIAS = -5.27                   # default used by tools/ when no calibration runs
TOWER_KERNEL = "conv_halo8_kernel<3, true>"   # symbol of the kernel the launcher picks for the head-tower layers (rtn_conv_halo8.hip)


def synth_images(torch, B, seed, device):
    g = torch.Generator().manual_seed(seed)
    raw = torch.clamp(torch.empty(B, CANVAS[0], CANVAS[1], 3).exponential_(1 / 12.0, generator=g) *
                      torch.rand(B, CANVAS[0], CANVAS[1], 3, generator=g), 0, 255).round()
    x = raw / 127.5 - 1.0                                      # model/utils.py:43-46
    return x.to(torch.bfloat16).to(device)


def conv_flops(op_desc, batch):
    """Algorithmic FLOPs of one conv launch: 2 * M * N * (true taps * Cin)."""
    d = op_desc
    k_true = 7 * 7 * 3 if (d.Crun == 32 and d.pix_stride == 4) else d.KH * d.KW * d.Crun
    m = sum(d.g[i].Hout * d.g[i].Wout for i in range(d.ngroups)) * batch
    return 2.0 * m * d.N * k_true


def conv_bytes(op_desc, batch):
    """Algorithmic HBM bytes of one conv launch: every input, weight, residual element read once, every output written once."""
    d = op_desc
    es = 4 if d.dtype == 1 else 2
    out_es = 4 if (d.flags & 0x10) else es
    total = d.N * d.KH * d.KW * d.Crun * es
    for i in range(d.ngroups):
        g = d.g[i]
        total += batch * g.Hin * g.Win * d.pix_stride * es
        total += batch * g.Hout * g.Wout * d.N * out_es
        if d.flags & 0x04:
            total += batch * g.Hout * g.Wout * d.N * es
        if d.flags & 0x08:
            total += batch * g.Hres * g.Wres * d.N * es
    return float(total)


SECONDARY_TIMEOUT_S = 240          # multi-GPU only: see the `secondary` block of main()


def pmc_traffic(n_conv_launches):
    """HBM bytes per step of the conv launches from the newest committed PMC summary (profiles/*pmc_traffic*.json: FETCH_SIZE x2
    + WRITE_SIZE, rocprofv3 --pmc in separate passes on this same bench command, see tools/pmc_traffic.py).  The counters are
    NOT collected in this run: the file is only quoted when it was collected on the same launch plan (conv launches per step),
    and the line says which file and commit.  Returns (bytes_per_step or None, note)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")))          # rN_vM_...: the name orders them
    if not files:
        return None, "no PMC summary committed"
    note = None
    for path in reversed(files):                     # the latest summary collected on THIS launch plan
        try:
            with open(path) as f:
                j = json.load(f)
            tot = j["conv_total"]
            launches = j.get("bench_launches_per_step")
        except (OSError, KeyError, ValueError) as e:
            note = note or "unreadable PMC summary %s: %s" % (os.path.basename(path), e)
            continue
        tag = "profiles/%s (commit %s, plan of %s conv launches/step, 2 x FETCH_SIZE + WRITE_SIZE)" % (
            os.path.basename(path), j.get("git_commit", "unrecorded"), launches)
        if launches is not None and int(launches) == int(n_conv_launches):
            break
        note = note or "stale: %s, this plan has %d" % (tag, n_conv_launches)
    else:
        return None, note
    return float(tot["hbm_bytes_per_step"]), tag


def pmc_traffic_train():
    """HBM bytes per TRAINING step (batch 16) from the newest committed profiles/*pmc_train_traffic*.json (tools/pmc_train.py: two
    rocprofv3 --pmc passes over `bench.py --mode train`, FETCH_SIZE x 2 + WRITE_SIZE summed over every kernel of a step)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_train_traffic*.json")))
    if not files:
        return None, "no PMC summary of the training step committed"
    try:
        with open(files[-1]) as f:
            j = json.load(f)
        return float(j["hbm_bytes_per_step"]), "profiles/%s (commit %s, %s steps of batch %s, 2 x FETCH_SIZE + WRITE_SIZE over all kernels)" % (
            os.path.basename(files[-1]), j.get("git_commit", "unrecorded"), j.get("steps"), j.get("batch"))
    except (OSError, KeyError, ValueError) as e:
        return None, "unreadable PMC summary %s: %s" % (os.path.basename(files[-1]), e)


def cpu_baseline(torch, state, threads):
    import numpy as np
    from oracle import ref_numpy as R
    from oracle.ref_net import RefNet
    torch.set_num_threads(threads)
    x = synth_images(torch, 1, 123, "cpu").float().numpy()
    net = RefNet(state, dtype=torch.float32)
    a32 = R.anchors_f32(CANVAS + (3,))
    t0 = time.perf_counter()
    n = 0
    while True:
        with torch.no_grad():
            reg, cls = net.forward(x)
        boxes = R.decode_boxes_f32(a32, reg.numpy()[0], CANVAS)
        R.filter_detections(boxes, cls.numpy()[0])
        n += 1
        dt = time.perf_counter() - t0
        if dt > 10.0 or n >= 16:                   # a bounded sample of about 10 s of host work (one image per second on 16 threads)
            break
    return {"value": n / dt, "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": "%d x (1 image 800x1333: fp32 torch-CPU forward + NumPy decode/NMS), %.1f s" % (n, dt)}


def train_algorithmic_bytes(eng, tr, bp, B, N, K, conv_bytes_fn):
    """Algorithmic HBM bytes of one training step: every tensor an op must read once and write once - forward launches,
    targets, loss forward / backward, every backward op (dY, X / filters, masks and residual gradients in; dX or dW out; the split
    slabs of the weight gradients are NOT algorithmic), sum of squares, Adam (w, m, v, g in; w, m, v and the bf16 filters out) and
    the data-gradient filter repack.  Returns (total, {group: bytes})."""
    es = 2
    plan = bp["plan"]
    groups = {"forward": 0.0, "targets+loss": 0.0, "dgrad": 0.0, "wgrad": 0.0, "other backward": 0.0, "optimizer": 0.0}
    eng.training = True
    try:
        ops = eng.active_ops(plan)
    finally:
        eng.training = False
    for op in ops:
        if op[0] in ("conv", "dual"):
            groups["forward"] += conv_bytes_fn(op)
        elif op[0] == "bneck":
            m = op[3]
            px = m["B"] * m["H"] * m["W"]
            groups["forward"] += 2.0 * px * (64 + 256 + 256 + 64 + (64 if m.get("tail") else 0))
        elif op[0] == "chain":
            m = op[3]
            groups["forward"] += 2.0 * m["pixels"] * ((6 * m["mid"] + m["proj_c"]) if m.get("proj_c") else 10 * m["mid"])
        elif op[0] == "stem":
            Bn, Hn, Wn = op[2]
            groups["forward"] += Bn * (CANVAS[0] * CANVAS[1] * 4 * 2 + ((Hn + 1) // 2) * ((Wn + 1) // 2) * 64 * (2 + 2 + 1))
    rows = B * N
    groups["targets+loss"] = rows * (5 + 2) * 4 + 2 * rows * (2 + 5 + K + 4) * 4 + rows * (K + 4) * 4
    for b in bp["bops"]:
        kind = b[0]
        if kind == "dgrad":
            groups["dgrad"] += conv_bytes_fn(("conv", b[1]))
        elif kind == "wgrad":
            d = b[1]
            t = d.N * d.KH * d.KW * d.Crun * 4
            for i in range(d.ngroups):
                g = d.g[i]
                t += B * g.Hin * g.Win * d.pix_stride * es + B * g.Hout * g.Wout * d.out_ld * es
            groups["wgrad"] += t
        elif kind == "padcast":
            groups["other backward"] += b[3] * (b[4] * 4 + b[5] * es)
        elif kind == "zins":
            Bn, Ho, Wo, Cc, Hs, Ws = b[3]
            groups["other backward"] += Bn * Cc * es * (Ho * Wo + Hs * Ws)
        elif kind == "upbwd":
            Bn, Hd, Wd, Hs, Ws, Cc = b[3]
            groups["other backward"] += Bn * Cc * es * (Hd * Wd + 2 * Hs * Ws)
        elif kind == "poolbwd":
            Bn, Hi, Wi, Cc = b[4]
            groups["other backward"] += Bn * Cc * (((Hi + 1) // 2) * ((Wi + 1) // 2) * (es + 1 + es) + Hi * Wi * es)
        elif kind == "bgrad":
            groups["other backward"] += b[2] * b[4] * es
    nparam = tr.NW + tr.NB
    groups["optimizer"] = nparam * 4 * 2 + nparam * (4 * 4 + 3 * 4) + tr.NW * es + 2 * tr.NW * es
    return sum(groups.values()), groups


TRAIN_BATCH = 16                   # BASELINE.json configs[2]: batch 16 per GPU
TRAIN_GFLOP_PER_IMAGE = 1248.4     # ~3x forward (dgrad + wgrad for every conv)


def measure_train(steps, warmup, torch, dist, E, Wt, rank, local_rank, world, device):
    """One step = forward + focal/smooth-L1 + backward + (bucketed RCCL all-reduce under DP) + clipnorm Adam on a batch of
    16 synthetic pages per GPU; anchor targets are produced on the device by rtn_anchor_targets inside the timed region.
    Same protocol as the inference line: `warmup` untimed steps, exactly `steps` timed steps between barrier + synchronize,
    max over ranks.  Returns the result object on rank 0 (None elsewhere)."""
    import ctypes as C
    import numpy as np
    T = importlib.import_module(PKG + ".trainer")
    L = importlib.import_module(PKG + "._lib")
    B = TRAIN_BATCH
    state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=-2.0, tame=True)
    eng = E.Engine("resnet50", 1, 9, dtype="bf16", device=local_rank)
    eng.load_state(state)
    tr = T.Trainer(eng, lr=1e-4, clipnorm=0.001, process_group=(dist.group.WORLD if dist is not None else None))
    x = synth_images(torch, B, 2000 + rank, device)
    cfg, N = E.make_anchor_cfg(CANVAS)
    rng = np.random.RandomState(100 + rank)
    gb, gc = np.zeros((B, 64, 4)), np.zeros(B, np.int32)
    for b in range(B):                                  # 1-6 boxes, w,h in [80,900]x[60,600]
        g = rng.randint(1, 7)
        w, h = rng.uniform(80, 900, g), rng.uniform(60, 600, g)
        x1, y1 = rng.uniform(0, CANVAS[1] - w), rng.uniform(0, CANVAS[0] - h)
        gb[b, :g] = np.stack([x1, y1, x1 + w, y1 + h], 1)
        gc[b] = g
    gbd, gld, gcd = torch.as_tensor(gb).to(device), torch.zeros(B, 64, dtype=torch.int32, device=device), torch.as_tensor(gc).to(device)
    hw = torch.as_tensor(np.tile(np.array(CANVAS, np.int32), (B, 1))).to(device)
    reg_t = torch.empty(B, N, 5, device=device)
    lab_t = torch.empty(B, N, 2, device=device)

    def step():
        eng._bind_stream()
        eng.h.check(L.lib.rtn_anchor_targets(eng.h.raw, C.byref(cfg), B, 1, gbd.data_ptr(), gld.data_ptr(), gcd.data_ptr(), hw.data_ptr(),
                                             0.4, 0.5, reg_t.data_ptr(), lab_t.data_ptr()))
        tr.forward_backward(x, reg_t, lab_t)
        tr.optimizer_step()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        return None
    value = world * B * steps / elapsed
    s = tr.norm_sums.cpu().numpy()
    achieved = value * TRAIN_GFLOP_PER_IMAGE / 1e3 / world
    # dominant kernel of the training step: the head-tower weight gradients (8 launches of the nine-tap window kernel, the largest
    # single share of the backward pass).  Measured live after the timed region: events on the launch stream around each launch,
    # launches one after another (untimed extra launches into the same gradient buffer, which the next step would zero anyway).
    dom = None
    try:
        bp = tr._bplan(B, CANVAS[0], CANVAS[1])
        tower = [(bi, b) for bi, b in enumerate(bp["bops"]) if b[0] == "wgrad" and str(b[3]).startswith(("pyramid_regression_", "pyramid_classification_"))
                 and bi in bp["rowinfo"]]
        if tower:
            eng._bind_stream()
            times = []
            for bi, b in tower:
                tab = bp["rowinfo"][bi]
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                ev[0].record()
                for _ in range(3):
                    eng.h.check(L.lib.rtn_conv2d_wgrad_prepared(eng.h.raw, C.byref(b[1]), b[2].data_ptr(), b[4].data_ptr() if b[4] is not None else None,
                                                                b[5] if b[4] is not None else 0, tab.data_ptr(), tab.numel()))
                ev[1].record()
                torch.cuda.synchronize()
                times.append(ev[0].elapsed_time(ev[1]) / 3)
            d0 = tower[0][1][1]
            fl = 2.0 * sum(d0.g[i].Hout * d0.g[i].Wout for i in range(d0.ngroups)) * B * d0.N * d0.KH * d0.KW * d0.Crun
            t_ms = sum(times) / len(times)
            impl = int(L.lib.rtn_debug_last_wgrad_impl(eng.h.raw))
            kname = {4: "conv_wgrad_win_kernel<0, false, true>", 2: "conv_wgrad_dma_kernel<4, 2, 8>"}.get(impl, "conv_wgrad_kernel")
            dom = {"kernel": kname + " + wgrad_finish_kernel (head-tower weight gradients: 3x3 256->256 over P3..P7, one grouped launch each)",
                   "launches_per_step": len(tower), "flop_per_launch": fl, "avg_launch_ms_solo": t_ms, "achieved_solo": fl / (t_ms * 1e-3) / 1e12,
                   "frac_solo": fl / (t_ms * 1e-3) / 1e12 / BF16_DENSE_PEAK_TFLOPS, "impl": impl,
                   "how": "events on the launch stream around 3 launches of each layer after the timed region (row-info table prebuilt, ordered slab reduction included)"}
    except Exception as e:              # the measurement beside the line may fail; the line itself may not
        dom = {"error": "%s: %s" % (type(e).__name__, e)}
    traffic, traffic_note = pmc_traffic_train()
    alg_bytes, alg_groups = None, None
    try:
        bp = tr._bplan(B, CANVAS[0], CANVAS[1])

        def cb(op):
            return conv_bytes(op[1], B) + ((B * op[3].Hin * op[3].Win * op[3].C + op[1].N * op[3].C) * 2.0 if op[0] == "dual" else 0.0)
        alg_bytes, alg_groups = train_algorithmic_bytes(eng, tr, bp, B, N, 1, cb)
    except Exception as e:              # beside the line, like `dominant`
        alg_groups = {"error": "%s: %s" % (type(e).__name__, e)}
    return {"metric": "images/sec RetinaNet R50-FPN 800x1333 training step", "value": value, "unit": "images/sec",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "ResNet50-FPN RetinaNet training step 800x1333 bf16 batch 16/GPU: targets + fwd + "
                                   "focal/smooth-L1 + bwd + clipnorm Adam (BASELINE.json configs[2]/[3])",
                       "batch_per_gpu": B, "positives_merged_batch": float(s[2]),
                       "parallelism": "dp%d (bucketed gradient all-reduce overlapped with backward)" % world},
            "roofline": {"bound": "mfma", "kernel": "conv fwd+dgrad+wgrad (whole step)", "achieved": achieved,
                         "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / BF16_DENSE_PEAK_TFLOPS,
                         "flop_per_step": TRAIN_GFLOP_PER_IMAGE * 1e9 * B, "traffic": traffic, "traffic_per": "step",
                         "traffic_source": traffic_note,
                         "algorithmic_bytes_per_step": alg_bytes, "algorithmic_bytes_by_group": alg_groups,
                         "hbm_floor_ms_at_6.3TBps": (alg_bytes / 6.3e12 * 1e3) if alg_bytes else None,
                         "traffic_over_algorithmic": (traffic / alg_bytes) if (traffic and alg_bytes) else None,
                         "dominant": dom}}


def bench_train(args, torch, dist, E, Wt, rank, local_rank, world, device):
    out = measure_train(args.steps, args.warmup, torch, dist, E, Wt, rank, local_rank, world, device)
    if rank == 0:
        out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the training-step measurement (the `secondary` object)")
    ap.add_argument("--secondary-steps", type=int, default=20)
    ap.add_argument("--in-flight", type=int, default=2,
                    help="batches the engine keeps in flight (Engine.in_flight): consecutive steps run on that many buffer sets, one HIP "
                         "stream each, and overlap on the device; 1 = one batch at a time with the graph's forks on side streams")
    ap.add_argument("--mode", choices=["infer", "train"], default="infer",
                    help="infer (default, BASELINE.json configs[1]) or train (configs[2]/[3]: batch 16/GPU training step)")
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` by itself: start the N ranks as a CHILD torch.distributed.run (one rank per GPU) before this
        # process has imported torch or touched the GPU, pass its output through (rank 0 prints the one JSON line) and return its
        # exit code.  The child sees WORLD_SIZE and takes the branch below.
        import socket
        import subprocess
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node equal to --gpus "
                         "(or run `python bench.py --gpus N` alone, which starts its own ranks)\n" % (args.gpus, world))
        sys.exit(2)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # One rank per GPU over RCCL ("nccl" on ROCm).  Rehearsal on a one-GPU box: RTN_BENCH_SHARE_GPU=1 puts every rank on
        # device 0 and RTN_DIST_BACKEND=gloo replaces RCCL (which refuses two ranks on one device).
        if os.environ.get("RTN_BENCH_SHARE_GPU") == "1":
            local_rank = 0
        backend = os.environ.get("RTN_DIST_BACKEND", "nccl")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    E = importlib.import_module(PKG + ".engine")
    Wt = importlib.import_module(PKG + ".weights")
    if args.mode == "train":
        return bench_train(args, torch, dist, E, Wt, rank, local_rank, world, device)
    x = synth_images(torch, BATCH, 1000 + rank, device)
    eng = E.Engine("resnet50", 1, 9, dtype="bf16", device=local_rank)
    # calibration (untimed): choose the classification bias so that CAND_FRACTION of the anchors clear the 0.05 score
    # threshold on this input, i.e. the NMS stage sees a trained detector's load rather than 0 or 200,700 candidates.
    state = Wt.init_state("resnet50", 1, 9, seed=0, randomize_bn=True, cls_bias=0.0, tame=True)
    eng.load_state(state)
    _, cls0 = eng.forward(x[:1])
    logit = torch.log(cls0 / (1 - cls0)).flatten().float()
    q = torch.quantile(logit[::7], 1.0 - CAND_FRACTION).item()
    cls_bias = math.log(0.05 / 0.95) - q
    state["pyramid_classification/bias"] = (state["pyramid_classification/bias"] * 0 + cls_bias).astype("float32")
    eng.load_state(state)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Throughput mode: `in_flight` batches at a time.  Every step is still one whole pass over one batch of 8 (stem pack -> backbone -> FPN ->
    # heads -> decode + NMS into that step's own output buffers); consecutive steps go to alternating buffer sets on separate HIP
    # streams and the timed region ends only when all K of them are done (join + barrier + synchronize).
    if args.in_flight < 1:
        ap.error("--in-flight must be >= 1")
    eng.in_flight = args.in_flight
    for _ in range(max(args.warmup, 0) if args.in_flight == 1 else 0):
        eng.detect(x)
    if args.in_flight > 1:                               # the buffer sets are built outside the W warm-up steps (a plan build is not a step)
        for _ in range(args.in_flight):
            eng.detect(x)
        eng.join()
        torch.cuda.synchronize()
        for _ in range(args.warmup):
            eng.detect(x)
        eng.join()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        boxes, scores, labels = eng.detect(x)
    eng.join()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    n_gpus = world
    images = n_gpus * BATCH * args.steps
    value = images / elapsed

    one_ms = None
    if args.in_flight > 1:                               # beside the line: the same steps one batch at a time (latency of a batch)
        eng.in_flight = 1
        for _ in range(3):
            eng.detect(x)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            eng.detect(x)
        torch.cuda.synchronize()
        one_ms = (time.perf_counter() - t1) / 10 * 1e3
    out = None
    if rank == 0:
        # ---- roofline.  `achieved` comes from the TIMED region (algorithmic conv FLOPs of a step / ms_per_step).  Beside it:
        # events around every launch of extra (untimed) steps, all launches on ONE stream one after another - the serial sum and
        # the solo duration of the dominant kernel (the head-tower layers: the largest share of GPU time).
        plan = eng._plan(BATCH, CANVAS[0], CANVAS[1])
        active = eng.active_ops(plan)
        conv_ops = [op for op in active if op[0] in ("conv", "dual", "bneck", "chain")]
        fused_stem = any(op[0] == "stem" for op in active)          # conv1 + ReLU + pool1 in one kernel: its MFMA work is conv1's
        stem_conv = [op for op in plan["ops"] if op[0] == "conv" and op[2] == "conv1"][0]
        # a "dual" launch is branch2c with the projection shortcut appended along K: the same FLOPs as the two layers it replaces
        def op_flops(op):
            if op[0] == "bneck":             # branch2b (3x3, 64 -> 64) + branch2c (64 -> 256) [+ the next branch2a (256 -> 64)]
                m = op[3]
                return 2.0 * m["B"] * m["H"] * m["W"] * (576 * 64 + 64 * 256 + 256 * 64 * (int(bool(m["tail"])) + int(bool(m.get("proj")))))
            if op[0] == "chain":             # branch2c (mid -> 4 mid) + the next branch2a (4 mid -> mid)
                return 2.0 * op[3]["pixels"] * (2 * op[3]["mid"] + op[3].get("proj_c", 0)) * 4 * op[3]["mid"]      # proj_c: + the projection shortcut
            return conv_flops(op[1], BATCH) + (2.0 * BATCH * op[1].g[0].Hout * op[1].g[0].Wout * op[1].N * op[3].C if op[0] == "dual" else 0.0)
        flops_step = sum(op_flops(op) for op in conv_ops) + (conv_flops(stem_conv[1], BATCH) if fused_stem else 0.0)
        stem_op = [op for op in active if op[0] == "stem"]
        stem_2a = [op for op in plan["ops"] if op[0] == "conv" and op[2] == "res2a_branch2a"] if (stem_op and stem_op[0][7] is not None) else []
        if stem_2a:                                                    # res2a_branch2a runs inside the stem kernel: its FLOPs still count
            flops_step += conv_flops(stem_2a[0][1], BATCH)
        ms_per_step = 1e3 * elapsed / args.steps
        achieved = flops_step / (ms_per_step * 1e-3) / 1e12
        reps = 3
        per_op_ms = eng.profile_ops(x, reps=reps)
        serial_conv_ms = sum(ms for kind, ms in per_op_ms if kind in ("conv", "stem", "dual", "bneck", "chain")) / reps
        serial_other_ms = sum(ms for kind, ms in per_op_ms if kind not in ("conv", "stem", "dual", "bneck", "chain")) / reps
        tower = [(op, ms / reps) for op, (kind, ms) in zip(active, per_op_ms[:len(active)])
                 if op[0] == "conv" and op[2].startswith(("pyramid_regression_", "pyramid_classification_"))]
        dom = None
        if tower:
            t_ms = sum(ms for _, ms in tower) / len(tower)
            t_fl = sum(op_flops(op) for op, _ in tower) / len(tower)
            dom = {"kernel": "%s (head-tower layers: 3x3 256->256 + ReLU over P3..P7, one grouped launch each)" % TOWER_KERNEL,
                   "launches_per_step": len(tower), "flop_per_launch": t_fl, "avg_launch_ms_solo": t_ms,
                   "achieved_solo": t_fl / (t_ms * 1e-3) / 1e12, "frac_solo": t_fl / (t_ms * 1e-3) / 1e12 / BF16_DENSE_PEAK_TFLOPS,
                   "share_of_serial_conv_time": len(tower) * t_ms / serial_conv_ms,
                   "bare_loop_ceiling": {"tflops": 1690.0, "what": "bf16 16x16x32 MFMAs + this kernel's fragment reads (6 ds_read_b128 per 16 MFMAs) "
                                         "on random operands, no staging / barriers / epilogue; register-only loop 1992 TF/s at 1.97 GHz",
                                         "source": "profiles/r3_micro_mfma_rate.txt (tools/micro/mfma_rate.hip, not measured in this run)"},
                   "how": "events on the launch stream around each launch, launches one after another, %d passes" % reps}
        ncand = int((plan["classification"] > 0.05).sum().item())
        def op_bytes(op):
            if op[0] == "bneck":             # a_in + shortcut in, x_out (+ a_out) out, the three filters
                m = op[3]
                px = m["B"] * m["H"] * m["W"]
                if m.get("proj"):            # a_in + the block input in, x_out (+ a_out) out, branch2b + the K-concatenated [branch2c | branch1] filters
                    return 2.0 * (px * (64 + 64 + 256 + (64 if m["tail"] else 0)) + 576 * 64 + 128 * 256 + (256 * 64 if m["tail"] else 0))
                return 2.0 * (px * (64 + 256 + 256 + (64 if m["tail"] else 0)) + 576 * 64 + 64 * 256 + (256 * 64 if m["tail"] else 0))
            if op[0] == "chain":             # branch2b's output + shortcut in, x_out + a_out out, the two filters
                m = op[3]
                if m.get("proj_c"):          # branch2b's output + the sampled block input in, x_out + a_out out, the filters
                    return 2.0 * (m["pixels"] * (6 * m["mid"] + m["proj_c"]) + 4 * m["mid"] * (2 * m["mid"] + m["proj_c"]))
                return 2.0 * (m["pixels"] * 10 * m["mid"] + 8 * m["mid"] * m["mid"])
            return conv_bytes(op[1], BATCH) + ((BATCH * op[3].Hin * op[3].Win * op[3].C + op[1].N * op[3].C) * 2.0 if op[0] == "dual" else 0.0)
        bytes_step = sum(op_bytes(op) for op in conv_ops)
        n_launch = len(conv_ops) + (1 if fused_stem else 0)
        if fused_stem:                                                 # image in (bf16, 3 ch), pooled tensor out, filters
            H1, W1 = (CANVAS[0] - 1) // 2 + 1, (CANVAS[1] - 1) // 2 + 1
            bytes_step += BATCH * (CANVAS[0] * CANVAS[1] * 3 * 2 + ((H1 + 1) // 2) * ((W1 + 1) // 2) * 64 * 2) + 64 * 256 * 2
            if stem_2a:                                                # + the branch2a tensor out, its filters in
                bytes_step += BATCH * ((H1 + 1) // 2) * ((W1 + 1) // 2) * 64 * 2 + 64 * 64 * 2
        traffic, traffic_note = pmc_traffic(n_launch)
        roofline = {"bound": "mfma", "kernel": "all conv launches of a step (implicit-GEMM MFMA kernels, bf16)",
                    "achieved": achieved, "peak": BF16_DENSE_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": achieved / BF16_DENSE_PEAK_TFLOPS,
                    "achieved_from": "flop_per_step / ms_per_step of the timed region",
                    "flop_per_step": flops_step, "launches_per_step": n_launch,
                    "traffic": traffic, "traffic_per": "step", "traffic_source": traffic_note,
                    "algorithmic_bytes_per_step": bytes_step,
                    "hbm_floor_ms_at_6.3TBps": bytes_step / 6.3e12 * 1e3,
                    "dominant": dom,
                    "serial_conv_ms_per_step": serial_conv_ms, "serial_non_conv_ms_per_step": serial_other_ms,
                    "serial_note": "per-launch event sums with every launch on one stream; larger than ms_per_step because the "
                                   "timed region runs the graph's forks on side streams"}
        out = {"metric": "images/sec RetinaNet R50-FPN 800x1333 inference", "value": value, "unit": "images/sec",
               "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": "ResNet50-FPN RetinaNet inference 800x1333 bf16 batch 8/GPU incl. decode+NMS "
                                      "(BASELINE.json configs[1])",
                          "batch_per_gpu": BATCH, "canvas": list(CANVAS), "anchors_per_image": plan["N"],
                          "candidates_above_0.05_per_image": ncand / BATCH, "parallelism": "dp%d (images sharded, no collective)" % n_gpus,
                          "batches_in_flight": args.in_flight, "ms_per_step_one_batch_at_a_time": one_ms,
                          "gflop_per_image_survey": GFLOP_PER_IMAGE, "gflop_per_image_counted": flops_step / BATCH / 1e9},
               "roofline": roofline}
        if not args.no_cpu_baseline and n_gpus == 1:
            threads = min(os.cpu_count() or 1, 16)
            out["cpu_baseline"] = cpu_baseline(torch, state, threads)
        else:
            out["cpu_baseline"] = None
    # ---- secondary: the training step (BASELINE.json configs[2]; configs[3] when the driver launches N ranks), every rank takes part
    if not args.no_secondary:
        del eng
        torch.cuda.empty_cache()
        # Under N ranks the training step holds collectives.  A rank that dies or diverges there would leave the others waiting in
        # RCCL, and the inference line - measured and complete at this point - would never be printed: a timer prints it with the
        # failure recorded and ends the process if the secondary measurement has not returned in time.
        timer = None
        if world > 1:
            import threading

            def give_up():
                if rank == 0:
                    out["secondary"] = {"error": "training-step measurement did not return within %d s (collective stalled?)" % SECONDARY_TIMEOUT_S}
                    print(json.dumps(out), flush=True)
                os._exit(0 if rank == 0 else 3)
            timer = threading.Timer(SECONDARY_TIMEOUT_S, give_up)
            timer.daemon = True
            timer.start()
        try:
            sec = measure_train(args.secondary_steps, min(args.warmup, 3), torch, dist, E, Wt, rank, local_rank, world, device)
        except Exception as e:          # the inference line must survive a failure here; the failure is reported, not hidden
            sec = {"error": "%s: %s" % (type(e).__name__, e)}
        if timer is not None:
            timer.cancel()
        if rank == 0:
            out["secondary"] = sec
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
