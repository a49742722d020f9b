"""Training step of the RetinaNet path on librtn.so: forward -> focal + smooth-L1 -> backward -> clipnorm Adam.

Mirrors what the reference gets from Keras (RetinaNet.py:125-131, 280-291):
  compile(loss={'regression': smooth_l1(), 'classification': focal()}, optimizer=Adam(lr=1e-4, clipnorm=0.001))
  -> train_on_batch(inputs, [regression_batch (B,N,5), labels_batch (B,N,K+1)])
  * total loss = smooth_l1 + focal with unit weights, each normalised by max(1, #positive anchors of the MERGED batch)
    (model/losses.py:39-44,87-90; under data parallelism the counts are all-reduced first);
  * every conv kernel and every FPN/head bias trains; BatchNorm is frozen (freeze_bn=True): its scale is folded into the
    forward weights, so the gradient w.r.t. the Keras kernel is fold[n] * dL/dW_folded and Adam runs on the unfolded
    master copy;
  * clipnorm: global-norm clipping by default (standalone Keras 2.x semantics, what the reference's `import keras` resolves to);
    `global_clip=False` clips every gradient tensor by its own norm (tf.keras / Keras >= 2.4 semantics).

The backward graph is derived from the forward op list of engine.Engine._plan: every conv gets a wgrad (+ bias grad) and a
dgrad per input; ReLU, residual adds, UpsampleLike+Add, C6_relu, max-pool and the stride-2 convs are handled by epilogue
flags of the dgrad launch or by the small kernels of rtn_backward.hip.  All arithmetic is in the HIP library; PyTorch
provides memory, the stream and (optionally) torch.distributed for the gradient all-reduce.
"""
import ctypes as C

import numpy as np
import os
import weakref

import torch

from . import _lib as L
from . import weights as Wt
from . import parallel as Par


def _ceil128(v):
    return -(-v // 128) * 128


class Trainer:
    WG_LANES = 3          # side streams a weight gradient may go to
    def __init__(self, engine, lr=1e-4, clipnorm=0.001, beta1=0.9, beta2=0.999, eps=1e-7, alpha=0.25, gamma=2.0, sigma=3.0,
                 process_group=None, global_clip=True):
        self.eng = engine
        self.global_clip = bool(global_clip)
        self.lr, self.clipnorm, self.b1, self.b2, self.eps = lr, clipnorm, beta1, beta2, eps
        self.alpha, self.gamma, self.sigma = alpha, gamma, sigma
        self.pg = process_group
        self.wgrad_lane = os.environ.get("RTN_WGRAD_LANE", "1") != "0"   # weight gradients on side HIP streams
        self.wgrad_lanes = max(1, min(self.WG_LANES, int(os.environ.get("RTN_WGRAD_LANES", "3"))))
        self._wg_stream = None
        self.record_impls = False      # tests: {(kind, layer): kernel code} of the last backward (rtn_debug_last_*_impl)
        self.impls = {}
        # a backward plan holds gradient buffers, descriptors into the forward plan's activations and every weight-gradient workspace
        # (row-info table + split slabs: GBs per canvas): it lives exactly as long as the engine's plan of the same canvas
        engine.on_plan_evict.append(weakref.WeakMethod(self._drop_bplan))
        self._bind_engine_state()

    def _drop_bplan(self, plan_key):
        if len(plan_key) == 4:                              # (B, H, W, fp8): the one-batch buffer set the backward plan points into
            self.bplans.pop(tuple(plan_key[:3]), None)

    def _bind_engine_state(self):
        """Everything derived from the engine's loaded state: the f32 master copy, fold / gradient scales, Adam moments, dgrad
        weights, the backward plans (which hold descriptors into the forward plan's activation buffers) and the gradient
        bucketer.  Engine.load_state() reallocates the flat weights and drops its plans, so a Trainer that lived through it
        starts over from the newly loaded weights with fresh optimizer state - exactly what a new Trainer would hold."""
        self._epoch = self.eng.load_epoch
        self.step_count = 0
        self.bplans = {}
        self._init_params()
        self.bucketer = None
        if self.pg is not None:
            segs = [(name, lo["woff"], lo["woff"] + lo["rows"] * lo["K"]) for name, lo in self.eng.layout.items()]
            segs.append(("__biases__", self.NW, self.NW + self.NB))           # FPN/head biases: one last segment
            self.bucketer = Par.GradBucketer(self.grad, segs, group=self.pg)

    def _check_engine_state(self):
        if self._epoch != self.eng.load_epoch:
            self._bind_engine_state()

    # ------------------------------------------------------------------ parameters
    def _init_params(self):
        eng, dev, st = self.eng, self.eng.device, self.eng.state
        NW, NB = eng.wflat.numel(), eng.bflat.numel()
        self.NW, self.NB = NW, NB
        master = torch.zeros(NW + NB, dtype=torch.float32)
        gscale = torch.zeros(NW + NB, dtype=torch.float32)
        fold = torch.ones(NW + NB, dtype=torch.float32)
        for name, lo in eng.layout.items():
            rows, K, cout = lo["rows"], lo["K"], lo["cout"]
            pack = Wt.pack_stem if name == "conv1" else Wt.pack_conv
            wu, _ = pack(st[name + "/kernel"], None, None, torch.float32, "cpu")          # unfolded kernel, same layout
            master[lo["woff"]:lo["woff"] + rows * K] = wu.reshape(-1)
            scale = torch.ones(rows)
            if lo["bn"] is not None:
                g_, v_ = [torch.as_tensor(np.asarray(st[lo["bn"] + s]), dtype=torch.float32) for s in ("/gamma", "/moving_variance")]
                scale[:cout] = g_ / torch.sqrt(v_ + Wt.BN_EPS)
            f = scale.view(rows, 1).expand(rows, K).clone()
            gs = f.clone()
            gs[cout:] = 0.0
            if name == "conv1":                      # structural zeros of the packed stem never train
                live = torch.zeros(rows, 8, 8, 4)
                live[:cout, :7, :7, :3] = 1.0
                gs = gs * live.reshape(rows, K)
            fold[lo["woff"]:lo["woff"] + rows * K] = f.reshape(-1)
            gscale[lo["woff"]:lo["woff"] + rows * K] = gs.reshape(-1)
            # bias slot: trainable only for layers that own a Keras bias
            bo = NW + lo["boff"]
            master[bo:bo + rows] = eng.bflat[lo["boff"]:lo["boff"] + rows].cpu()
            if lo["has_bias"]:
                gscale[bo:bo + cout] = 1.0
        self.master = master.to(dev)
        self.gscale = gscale.to(dev)
        self.fold = fold.to(dev)
        self.m = torch.zeros(NW + NB, dtype=torch.float32, device=dev)
        self.v = torch.zeros(NW + NB, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(NW + NB, dtype=torch.float32, device=dev)
        self.sumsq = torch.zeros(1, dtype=torch.float64, device=dev)
        # per-tensor clipping: one segment per conv kernel and per bias vector (a Keras weight tensor each), in flat order
        begins = [lo["woff"] for lo in eng.layout.values()] + [NW + lo["boff"] for lo in eng.layout.values()] + [NW + NB]
        assert begins == sorted(begins)
        self.seg_begin = torch.tensor(begins, dtype=torch.int64, device=dev)
        self.nseg = len(begins) - 1
        self.sumsq_seg = torch.zeros(self.nseg, dtype=torch.float64, device=dev)
        self.ss_ws = torch.empty(L.lib.rtn_sumsq_workspace_bytes(), dtype=torch.uint8, device=dev)
        self.loss_sums = torch.zeros(4, dtype=torch.float64, device=dev)
        # dgrad weights (re-packed from the forward weights after every optimizer step)
        self.wd = {}
        for name, lo in eng.layout.items():
            if name == "conv1":
                continue
            crun = self._dy_channels(name)
            self.wd[name] = torch.empty(_ceil128(lo["cin"]), lo["kh"] * lo["kw"] * crun, dtype=eng.tdt, device=dev)
        self._pack_table = None
        self._repack_dgrad()

    def _dy_channels(self, name):
        """Channels of the dY tensor a layer's backward reads: Cout, or the padded width for the skinny head outputs."""
        cout = self.eng.layout[name]["cout"]
        if name in ("pyramid_regression", "pyramid_classification"):
            cp = 64
            while cp < cout:
                cp *= 2
            return cp
        return cout

    def _repack_dgrad(self):
        """Every layer's dgrad weights from the (rewritten) forward weights, one launch: rtn_pack_dgrad_weights_multi."""
        eng = self.eng
        eng._bind_stream()
        if self._pack_table is None or self._pack_table[3] != eng.wflat.data_ptr():      # (re)built when the weights were reallocated
            rows, total = [], 0
            for name, wd in self.wd.items():
                lo = eng.layout[name]
                rows.append([eng.w[name][0].data_ptr(), wd.data_ptr(), lo["cout"], lo["kh"], lo["kw"], lo["cin"], self._dy_channels(name),
                             wd.shape[0], total, 0])
                total += wd.numel()
            self._pack_table = (torch.tensor(rows, dtype=torch.int64, device=eng.device), len(rows), total, eng.wflat.data_ptr())
        table, n, total, _ = self._pack_table
        eng.h.check(L.lib.rtn_pack_dgrad_weights_multi(eng.h.raw, table.data_ptr(), n, total, eng.rdt))

    def grad_views(self, name):
        """(dW [rows][K] f32, db [rows] f32) views of the flat gradient buffer for one layer."""
        lo = self.eng.layout[name]
        return (self.grad[lo["woff"]:lo["woff"] + lo["rows"] * lo["K"]].view(lo["rows"], lo["K"]),
                self.grad[self.NW + lo["boff"]:self.NW + lo["boff"] + lo["rows"]])

    # ------------------------------------------------------------------ backward plan
    def _bplan(self, B, H, W):
        key = (B, H, W)
        eng = self.eng
        if key in self.bplans:
            eng._plan(B, H, W)                             # marks the canvas as recently used in the engine's cache
            return self.bplans[key]
        plan = eng._plan(B, H, W)                          # may evict the least recently used canvas: _drop_bplan() follows
        dev, tdt, rdt = eng.device, eng.tdt, eng.rdt
        ops = plan["ops"]
        keep, bops = [], []
        tid = id

        # producers / consumers
        prod, ncons, relu_src = {}, {}, {}
        for op in ops:
            if op[0] == "conv":
                me = op[3]
                for y in me["ys"]:
                    prod[tid(y)] = me
                for x in me["xs"]:
                    ncons[tid(x)] = ncons.get(tid(x), 0) + 1
                for r in me["res"]:
                    if r is not None:
                        ncons[tid(r)] = ncons.get(tid(r), 0) + 1
            elif op[0] == "pool":
                ncons[tid(op[1])] = ncons.get(tid(op[1]), 0) + 1
                prod[tid(op[2])] = {"relu": False, "name": "pool1"}
            elif op[0] == "relu":
                relu_src[tid(op[2])] = op[1]                       # P6r -> P6
        for t_, src in relu_src.items():                            # a consumer of relu(S) counts as a consumer of S
            ncons[tid(src)] = ncons.get(tid(src), 0) + ncons.pop(t_, 0)

        grads, gstate, done = {}, {}, {}

        def gbuf(t):
            if tid(t) not in grads:
                grads[tid(t)] = torch.zeros_like(t)                 # zeros: scatter targets and never-touched pixels stay 0
                keep.append(grads[tid(t)])
            return grads[tid(t)]

        cells_total = sum(plan["cfg"].H[i] * plan["cfg"].W[i] for i in range(plan["cfg"].nlevels))
        N = plan["N"]
        d_reg = torch.zeros(B, N, 4, dtype=torch.float32, device=dev)
        d_cls = torch.zeros(B, N, eng.K, dtype=torch.float32, device=dev)
        cp_reg, cp_cls = self._dy_channels("pyramid_regression"), self._dy_channels("pyramid_classification")
        dyp_reg = torch.zeros(B, cells_total, cp_reg, dtype=tdt, device=dev)
        dyp_cls = torch.zeros(B, cells_total, cp_cls, dtype=tdt, device=dev)
        keep += [d_reg, d_cls, dyp_reg, dyp_cls]
        grads[tid(plan["regression"])] = dyp_reg
        grads[tid(plan["classification"])] = dyp_cls
        bops.append(("padcast", d_reg, dyp_reg, B * cells_total, eng.A * 4, cp_reg))
        bops.append(("padcast", d_cls, dyp_cls, B * cells_total, eng.A * eng.K, cp_cls))

        def group_fwd_geom(g, x):
            g.in_, g.in_elems = x.data_ptr(), x.numel()

        max_ws = 0
        for op in reversed(ops):
            kind = op[0]
            if kind == "conv":
                d_f, name, me = op[1], op[2], op[3]
                lo = eng.layout[name]
                head_out = name in ("pyramid_regression", "pyramid_classification")
                ng = d_f.ngroups
                crun = self._dy_channels(name)
                def grad_of(t):                      # a tensor whose only consumer is a residual add: its gradient IS that sum's
                    if tid(t) in grads:
                        return grads[tid(t)]
                    st_ = gstate.get(tid(t))
                    return st_[1] if isinstance(st_, tuple) else None
                dys = [grad_of(y) for y in me["ys"]]
                if any(dy is None for dy in dys):
                    raise RuntimeError("no gradient reached the output of %s" % name)
                dW, db = self.grad_views(name)
                # ---------- wgrad: forward geometry, `out` re-pointed at dY
                dw = L.ConvDesc()
                C.memmove(C.byref(dw), C.byref(d_f), C.sizeof(L.ConvDesc))
                dw.flags, dw.w, dw.bias = 0, None, None
                dw.N = crun
                dw.out_ld = crun
                cell_off = 0
                for gi in range(ng):
                    g = dw.g[gi]
                    dy = dys[gi]
                    g.out, g.out_elems = dy.data_ptr(), dy.numel()
                    g.res, g.res_elems = None, 0
                    cells = g.Hout * g.Wout
                    if head_out:
                        g.out_img_stride, g.out_off = cells_total * crun, cell_off * crun
                        cell_off += cells
                    else:
                        g.out_img_stride, g.out_off = cells * crun, 0
                wsb = L.lib.rtn_conv2d_wgrad_workspace_bytes(C.byref(dw))
                max_ws = max(max_ws, wsb)
                bops.append(("wgrad", dw, dW, name, db if lo["has_bias"] else None, lo["cout"]))     # bias gradient fused
                # ---------- residual inputs of the forward epilogue
                for gi in range(ng):
                    r = me["res"][gi]
                    if r is None:
                        continue
                    done[tid(r)] = done.get(tid(r), 0) + 1
                    if me["flags"] & L.CONV_RES_SAME:
                        if gstate.get(tid(r)) is not None:
                            raise RuntimeError("identity gradient must be the first contribution (%s)" % name)
                        gstate[tid(r)] = ("alias", dys[gi])
                    else:                                            # UpsampleLike + Add
                        acc = 1 if gstate.get(tid(r)) == "buf" else 0
                        bops.append(("upbwd", dys[gi], gbuf(r), (B, dys[gi].shape[1], dys[gi].shape[2], r.shape[1], r.shape[2], r.shape[3]), acc))
                        gstate[tid(r)] = "buf"
                # ---------- dgrad into every input
                if me.get("stem"):
                    continue
                dd = L.ConvDesc()
                dd.ngroups, dd.batch, dd.dtype = ng, B, rdt
                wd = self.wd[name]
                dd.w, dd.bias, dd.w_rows, dd.N = wd.data_ptr(), None, wd.shape[0], lo["cin"]
                dd.KH, dd.KW, dd.Crun, dd.pix_stride = lo["kh"], lo["kw"], crun, crun
                dd.sy = dd.sx = 1
                dd.out_ld = lo["cin"]
                stride, (pt, pl) = me["stride"], me["pad"]
                flags_all = None
                cell_off = 0
                for gi in range(ng):
                    x = me["xs"][gi]
                    target, pre_mask = x, False
                    if tid(x) in relu_src:                           # x = relu(S): differentiate straight into S
                        target, pre_mask = relu_src[tid(x)], True
                    done[tid(target)] = done.get(tid(target), 0) + 1
                    state = gstate.get(tid(target))
                    dy = dys[gi]
                    g = L.ConvGroup()
                    Hx, Wx = x.shape[1], x.shape[2]
                    Ho, Wo = d_f.g[gi].Hout, d_f.g[gi].Wout
                    flags = 0
                    if stride == 1:
                        src, Hs, Ws = dy, Ho, Wo
                        dd.pad_t, dd.pad_l = lo["kh"] - 1 - pt, lo["kw"] - 1 - pl
                        g.Hout, g.Wout = Hx, Wx
                        if head_out:
                            g.in_img_stride, in_off = cells_total * crun, cell_off * crun
                            cell_off += Ho * Wo
                        else:
                            g.in_img_stride, in_off = Ho * Wo * crun, 0
                        g.in_, g.in_elems = src.data_ptr() + in_off * src.element_size(), src.numel() - in_off
                        g.in_row_stride = Wo * crun
                    elif lo["kh"] == 1:                              # stride-2 1x1 'valid': scatter onto the stride grid
                        src, Hs, Ws = dy, Ho, Wo
                        dd.pad_t = dd.pad_l = 0
                        g.Hout, g.Wout = Ho, Wo
                        g.out_step, g.out_pix_w = stride, Wx
                        g.in_, g.in_elems = src.data_ptr(), src.numel()
                        g.in_img_stride, g.in_row_stride = Ho * Wo * crun, Wo * crun
                    else:                                            # stride-2 3x3: stride-1 conv over the zero-inserted dY
                        Hs, Ws = 2 * Ho - 1, 2 * Wo - 1
                        up = torch.zeros(B, Hs, Ws, crun, dtype=tdt, device=dev)
                        keep.append(up)
                        bops.append(("zins", dy, up, (B, Ho, Wo, crun, Hs, Ws)))
                        src = up
                        dd.pad_t, dd.pad_l = lo["kh"] - 1 - pt, lo["kw"] - 1 - pl
                        g.Hout, g.Wout = Hx, Wx
                        g.in_, g.in_elems = src.data_ptr(), src.numel()
                        g.in_img_stride, g.in_row_stride = Hs * Ws * crun, Ws * crun
                    g.Hin, g.Win = Hs, Ws
                    out = gbuf(target)
                    g.out, g.out_elems = out.data_ptr(), out.numel()
                    g.out_img_stride = out.numel() // B
                    if state is not None:
                        rs = out if state == "buf" else state[1]
                        flags |= L.CONV_RES_SAME
                        g.res, g.res_elems, g.res_img_stride, g.res_ld = rs.data_ptr(), rs.numel(), rs.numel() // B, rs.shape[-1]
                    needs_relu = pre_mask or prod.get(tid(target), {}).get("relu", False)
                    if needs_relu:
                        flags |= L.CONV_RELU_MASK
                        # an aliased identity gradient is not yet masked by this tensor's ReLU: mask the sum; otherwise
                        # mask this contribution alone (the accumulated part was masked when it was written)
                        if not (state is not None and state != "buf"):
                            flags |= L.CONV_MASK_PRE
                        g.mask, g.mask_elems, g.mask_img_stride, g.mask_ld = target.data_ptr(), target.numel(), target.numel() // B, target.shape[-1]
                    gstate[tid(target)] = "buf"
                    if flags_all is None:
                        flags_all = flags
                    elif flags_all != flags:
                        raise RuntimeError("grouped dgrad of %s needs uniform epilogue flags" % name)
                    dd.g[gi] = g
                dd.flags = flags_all
                L.attach_conv_workspace(eng.h, dd)                   # caller-owned K-split slabs, one buffer per launch
                bops.append(("dgrad", dd, name))
            elif kind == "pool":
                x, y = op[1], op[2]
                dy = grads.get(tid(y))
                done[tid(x)] = done.get(tid(x), 0) + 1
                bops.append(("poolbwd", x, dy, gbuf(x), op[3], op[4], y))      # y: the pooled tensor (the fused stem's ReLU mask)
                gstate[tid(x)] = "buf"
        loss_ws = torch.empty(L.lib.rtn_retina_loss_workspace_bytes(B * N), dtype=torch.uint8, device=dev)
        bias_bops = [i for i, b in enumerate(bops) if b[0] == "wgrad" and b[4] is not None]
        # every weight-gradient op owns its workspace (row-info table + the slabs of its ordered split reduction: 130 MB for a head
        # layer at batch 16 x 800 x 1333, 3.4 GB over the 107 layers), allocated at its first launch in forward_backward: ops on
        # different lanes never share scratch, and the table is built once per layer
        bp = {"bops": bops, "keep": keep, "last_bias_bop": bias_bops[-1] if bias_bops else -1, "wgrad_ws_bytes": max_ws,
              "d_reg": d_reg, "d_cls": d_cls, "dyp_cls": dyp_cls, "loss_ws": loss_ws, "plan": plan, "rowinfo": {}}
        self.bplans[key] = bp
        return bp

    # ------------------------------------------------------------------ step
    # ------------------------------------------------------------------ backward lanes
    @staticmethod
    def _bop_io(b):
        """(pointers read, pointers written) of a backward op (accumulating outputs are both)."""
        kind = b[0]
        if kind == "padcast":
            return [b[1].data_ptr()], [b[2].data_ptr()]
        if kind == "zins":
            return [b[1].data_ptr()], [b[2].data_ptr()]
        if kind == "upbwd":
            return [b[1].data_ptr()] + ([b[2].data_ptr()] if b[4] else []), [b[2].data_ptr()]
        if kind == "poolbwd":
            return [b[1].data_ptr(), b[2].data_ptr(), b[5].data_ptr(), b[6].data_ptr()], [b[3].data_ptr()]
        d = b[1]
        if kind == "wgrad":
            reads = [p_ for i in range(d.ngroups) for p_ in (d.g[i].in_, d.g[i].out)]
            return reads, [b[2].data_ptr()] + ([b[4].data_ptr()] if b[4] is not None else [])
        if kind == "dgrad":
            reads, writes = [], []
            for i in range(d.ngroups):
                g = d.g[i]
                reads.append(g.in_)
                if d.flags & L.CONV_RELU_MASK:
                    reads.append(g.mask)
                if d.flags & (L.CONV_RES_SAME | L.CONV_RES_UPSAMPLE):
                    reads.append(g.res)
                writes.append(g.out)
            return reads, writes
        raise RuntimeError(kind)

    def _bschedule(self, bops, nwg, dyp_cls):
        """Lane of every backward op and the cross-lane events it must wait for.  Lane 0: the data-gradient chain; lanes
        1..nwg: weight / bias gradients (independent of each other: round-robin); lane nwg+1: the data gradients of the
        classification tower, which only meet the rest of the graph where the pyramid gradients are summed.  Hazards are
        tracked per buffer: read-after-write, write-after-write and write-after-read (gradient buffers are accumulated in
        place by several ops)."""
        lanes, turn = [], 0
        extra = os.environ.get("RTN_BWD_EXTRA_LANE", "0") != "0"
        for b in bops:
            if b[0] in ("wgrad", "bgrad"):
                lanes.append(1 + turn % nwg)
                turn += 1
            elif (b[0] == "dgrad" and str(b[2]).startswith("pyramid_classification")) or (b[0] == "padcast" and b[2] is dyp_cls):
                lanes.append(nwg + 1)
            elif extra and b[0] == "dgrad" and (str(b[2]).endswith("_branch1") or str(b[2]) in ("P6", "P7", "P5", "P4")):
                lanes.append(nwg + 2)
            else:
                lanes.append(0)
        writer, readers, waits = {}, {}, []
        for i, b in enumerate(bops):
            reads, writes = Trainer._bop_io(b)
            reads = [p_ for p_ in reads if p_]
            writes = [p_ for p_ in writes if p_]
            deps = set()
            for ptr in reads:
                j = writer.get(ptr)
                if j is not None and lanes[j] != lanes[i]:
                    deps.add(j)
            for ptr in writes:
                j = writer.get(ptr)
                if j is not None and lanes[j] != lanes[i]:
                    deps.add(j)
                for j in readers.get(ptr, ()):
                    if lanes[j] != lanes[i]:
                        deps.add(j)
            latest = {}
            for j in deps:
                latest[lanes[j]] = max(latest.get(lanes[j], -1), j)
            waits.append(sorted(latest.values()))
            for ptr in reads:
                readers.setdefault(ptr, []).append(i)
            for ptr in writes:
                writer[ptr] = i
                readers[ptr] = []
        events = set(j for w in waits for j in w)
        last = {}
        for i, ln in enumerate(lanes):
            last[ln] = i
        joins = sorted(i for ln, i in last.items() if ln != 0)
        events.update(joins)
        return {"lanes": lanes, "waits": waits, "events": events, "joins": joins, "nlanes": max(lanes) + 1}

    def forward_backward(self, images, regression_batch, labels_batch):
        """Forward, loss and backward; leaves dL/dparams in self.grad (flat f32) and returns the device tensor
        loss_sums = [sum focal terms, sum smooth-L1 terms, #positives (labels), #positives (regression)] of THIS rank."""
        self._check_engine_state()
        eng, lib = self.eng, L.lib
        h = eng.h
        B, H, W, _ = images.shape
        eng.training = True
        try:
            reg, cls = eng.forward(images)
        finally:
            eng.training = False
        bp = self._bplan(B, H, W)
        N, K = bp["plan"]["N"], eng.K
        rows = B * N
        eng._bind_stream()
        h.check(lib.rtn_retina_loss_fwd(h.raw, rows, K, labels_batch.data_ptr(), regression_batch.data_ptr(), cls.data_ptr(),
                                        reg.data_ptr(), self.alpha, self.gamma, self.sigma, self.loss_sums.data_ptr(),
                                        bp["loss_ws"].data_ptr(), bp["loss_ws"].numel()))
        norm = self.loss_sums
        if self.pg is not None:                      # merged-batch normaliser (multi_gpu_model semantics)
            norm = Par.allreduce_loss_sums(self.loss_sums, self.pg)
        self.norm_sums = norm
        h.check(lib.rtn_retina_loss_bwd_dev(h.raw, rows, K, labels_batch.data_ptr(), regression_batch.data_ptr(), cls.data_ptr(),
                                            reg.data_ptr(), self.alpha, self.gamma, self.sigma, norm.data_ptr(), 1,
                                            bp["d_cls"].data_ptr(), bp["d_reg"].data_ptr()))
        self.grad.zero_()
        # Lanes (see _bschedule): data-gradient chain on the launch stream; weight / bias gradients round-robin over side streams;
        # the classification tower's data gradients on one more.  Events follow the per-buffer hazards.
        main = torch.cuda.current_stream(eng.device)
        lane_on = self.wgrad_lane
        nwg = self.wgrad_lanes        # also under DP: the bucketer records one event per (bucket, lane) and waits for all of them
        if lane_on:
            key = ("bsched", nwg)
            if key not in bp:
                sch = self._bschedule(bp["bops"], nwg, bp["dyp_cls"])
                sch["ev"] = {i: torch.cuda.Event() for i in sch["events"]}
                bp[key] = sch
            sch = bp[key]
            if self._wg_stream is None or len(self._wg_stream) < sch["nlanes"] - 1:
                # lanes on hardware queues of their own as far as the runtime has them (Engine._concurrent_streams)
                self._wg_stream = eng._concurrent_streams(self.WG_LANES + 2, beside=(main,))
                self._wg_ev = torch.cuda.Event()
            streams = [main] + self._wg_stream[:sch["nlanes"] - 1]
            self._wg_ev.record(main)                      # side lanes start behind the loss backward and the gradient reset
            for st in streams[1:]:
                st.wait_event(self._wg_ev)
        for bi, b in enumerate(bp["bops"]):
            kind = b[0]
            on_side = False
            if lane_on:
                ln = sch["lanes"][bi]
                st = streams[ln]
                for j in sch["waits"][bi]:
                    st.wait_event(sch["ev"][j])
                on_side = ln != 0
                if on_side:
                    side = st
                    h.set_stream(st.cuda_stream)
            if kind == "wgrad":
                tab = bp["rowinfo"].get(bi)
                if tab is None:                       # the row-info table depends on the descriptor only: built once per layer
                    tab = torch.empty(L.lib.rtn_conv2d_wgrad_workspace_bytes(C.byref(b[1])), dtype=torch.uint8, device=eng.device)
                    h.check(lib.rtn_conv2d_wgrad_rowinfo(h.raw, C.byref(b[1]), tab.data_ptr(), tab.numel()))
                    bp["rowinfo"][bi] = tab
                h.check(lib.rtn_conv2d_wgrad_prepared(h.raw, C.byref(b[1]), b[2].data_ptr(), b[4].data_ptr() if b[4] is not None else None,
                                                      b[5] if b[4] is not None else 0, tab.data_ptr(), tab.numel()))
                if self.record_impls:
                    self.impls[("wgrad", b[3])] = lib.rtn_debug_last_wgrad_impl(h.raw)
                if self.bucketer is not None:         # this layer's weight gradient is enqueued: its bucket may go out
                    done_names = [b[3]] + (["__biases__"] if bi == bp["last_bias_bop"] else [])   # last fused bias gradient
                    if len(done_names) > 1 and lane_on:   # the bias gradients came from every weight-gradient lane: mark them all
                        for ln_ in range(1, nwg + 1):
                            with torch.cuda.stream(streams[ln_]):
                                self.bucketer.mark("__biases__")
                    for dn in done_names:
                        if on_side:
                            with torch.cuda.stream(side):     # the bucket's event must follow the kernels on THEIR stream
                                self.bucketer.layer_done(dn)
                        else:
                            self.bucketer.layer_done(dn)
            elif kind == "dgrad":
                h.check(lib.rtn_conv2d_dgrad(h.raw, C.byref(b[1])))
                if self.record_impls:
                    self.impls[("dgrad", b[2])] = lib.rtn_debug_last_conv_impl(h.raw)
            elif kind == "bgrad":
                h.check(lib.rtn_bias_grad(h.raw, b[1].data_ptr(), eng.rdt, b[2], b[3], b[4], b[5].data_ptr()))
            elif kind == "padcast":
                h.check(lib.rtn_pad_cast_rows(h.raw, b[1].data_ptr(), b[2].data_ptr(), eng.rdt, b[3], b[4], b[5]))
            elif kind == "zins":
                Bn, Ho, Wo, Cc, Hs, Ws = b[3]
                h.check(lib.rtn_zero_insert2(h.raw, b[1].data_ptr(), b[2].data_ptr(), eng.rdt, Bn, Ho, Wo, Cc, Hs, Ws))
            elif kind == "upbwd":
                Bn, Hd, Wd, Hs, Ws, Cc = b[3]
                h.check(lib.rtn_upsample_add_bwd(h.raw, b[1].data_ptr(), b[2].data_ptr(), eng.rdt, Bn, Hd, Wd, Hs, Ws, Cc, b[4]))
            elif kind == "poolbwd":
                Bn, Hi, Wi, Cc = b[4]
                # the fused stem (bf16 training forward) never writes conv1's output: the ReLU mask comes from the pooled tensor
                fused = eng.fuse_stem and eng.fuse_stem_train and eng.dtype == "bf16"
                h.check(lib.rtn_maxpool3x3s2_tfsame_bwd_idx(h.raw, b[2].data_ptr(), b[5].data_ptr(), (b[6] if fused else b[1]).data_ptr(),
                                                            b[3].data_ptr(), eng.rdt, Bn, Hi, Wi, Cc, 2 if fused else 1))
            else:
                raise RuntimeError(kind)
            if lane_on:
                if bi in sch["ev"]:
                    sch["ev"][bi].record(st)
                if on_side:
                    h.set_stream(main.cuda_stream)
        if lane_on:
            for j in sch["joins"]:
                main.wait_event(sch["ev"][j])
        return self.loss_sums

    def optimizer_step(self, lr=None):
        """Global-norm clip + Adam on the flat parameter vector, re-emission of the forward and dgrad weights."""
        if self._epoch != self.eng.load_epoch:
            raise RuntimeError("Engine.load_state() ran between forward_backward() and optimizer_step(): the gradient belongs to "
                               "the previous weights")
        eng, lib, h = self.eng, L.lib, self.eng.h
        eng.join()                                   # inference batches still in flight (Engine.in_flight > 1) read the weights rewritten below
        eng._bind_stream()
        if self.bucketer is not None:                # sum of per-rank gradients == gradient of the merged batch
            self.bucketer.finish()
        self.step_count += 1
        n = self.NW + self.NB
        h.check(lib.rtn_sumsq(h.raw, self.grad.data_ptr(), self.gscale.data_ptr(), n, self.sumsq.data_ptr(), self.ss_ws.data_ptr(),
                              self.ss_ws.numel()))
        lr = self.lr if lr is None else lr
        if not self.global_clip:
            h.check(lib.rtn_sumsq_segments(h.raw, self.grad.data_ptr(), self.gscale.data_ptr(), self.seg_begin.data_ptr(), self.nseg,
                                           self.sumsq_seg.data_ptr()))
        for lo_, cnt, wf, code in ((0, self.NW, eng.wflat, eng.rdt), (self.NW, self.NB, eng.bflat, L.RTN_F32)):
            off4 = lo_ * 4
            if not self.global_clip:
                h.check(lib.rtn_adam_clipnorm_step_segments(
                    h.raw, self.master.data_ptr() + off4, self.m.data_ptr() + off4, self.v.data_ptr() + off4, self.grad.data_ptr() + off4,
                    self.gscale.data_ptr() + off4, self.fold.data_ptr() + off4, wf.data_ptr(), code, cnt, self.step_count, lr, self.b1,
                    self.b2, self.eps, self.seg_begin.data_ptr(), self.nseg, self.sumsq_seg.data_ptr(), lo_, self.clipnorm, 1.0))
                continue
            h.check(lib.rtn_adam_clipnorm_step(h.raw, self.master.data_ptr() + off4, self.m.data_ptr() + off4, self.v.data_ptr() + off4,
                                               self.grad.data_ptr() + off4, self.gscale.data_ptr() + off4, self.fold.data_ptr() + off4,
                                               wf.data_ptr(), code, cnt, self.step_count, lr, self.b1, self.b2, self.eps,
                                               self.sumsq.data_ptr(), self.clipnorm, 1.0))
        self._repack_dgrad()
        eng.weights_version += 1                        # fused inference copies of the filters are stale now

    def train_on_batch(self, images, regression_batch, labels_batch, lr=None):
        """Keras-style step. Returns (total, regression_loss, classification_loss) as Python floats (one host sync)."""
        self.forward_backward(images, regression_batch, labels_batch)
        self.optimizer_step(lr)
        s = self.norm_sums.cpu().numpy()
        reg_loss = float(s[1] / max(1.0, s[3]))
        cls_loss = float(s[0] / max(1.0, s[2]))
        return reg_loss + cls_loss, reg_loss, cls_loss
