"""Executor of the RetinaNet graph on librtn.so: one ctypes call per fused layer.

Graph (what a forward pass IS in the reference):
  resnet_retinanet (model/defineModel.py:357-389) -> keras_resnet backbone C3..C5
  __create_pyramid_features (:170-205) -> P3..P7
  regression / classification submodels on every level, concatenated (:208-228, :255-265)
  retinanet_bbox (:296-353): Anchors -> RegressBoxes -> ClipBoxes -> FilterDetections.
PyTorch is used for device memory and the stream only; all arithmetic is in the HIP library.
Frozen BN, bias, ReLU, the residual add, UpsampleLike+Add and the sigmoid are fused into the
convolution epilogues; the five pyramid levels of a head layer run as ONE grouped launch.
"""
import ctypes as C

import numpy as np
import os
import torch

from . import _lib as L
from . import weights as Wt

_TORCH_DT = {"bf16": torch.bfloat16, "f32": torch.float32}
_RTN_DT = {"bf16": L.RTN_BF16, "f32": L.RTN_F32}
_SRC_DT = {torch.bfloat16: 0, torch.float32: 1, torch.uint8: 2}


def same_pad_before(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2


class AnchorParams:
    """model/anchors.py:7-33 values (sizes/strides per level, float32 ratios/scales)."""

    def __init__(self, sizes=(32, 64, 128, 256, 512), strides=(8, 16, 32, 64, 128),
                 ratios=None, scales=None):
        self.sizes = list(sizes)
        self.strides = list(strides)
        self.ratios = np.array([0.5, 1, 2], np.float32) if ratios is None else np.asarray(ratios)
        self.scales = (np.array([2 ** 0, 2 ** (1.0 / 3.0), 2 ** (2.0 / 3.0)], np.float32)
                       if scales is None else np.asarray(scales))

    def num_anchors(self):
        return len(self.ratios) * len(self.scales)


def make_anchor_cfg(image_hw, params=None, levels=(3, 4, 5, 6, 7)):
    """rtn_anchor_cfg_t for a canvas: guess_shapes (model/anchors.py:155-165) + base anchors."""
    params = params or AnchorParams()
    cfg = L.AnchorCfg()
    cfg.nlevels = len(levels)
    cfg.A = params.num_anchors()
    off = 0
    for i, lv in enumerate(levels):
        h = (int(image_hw[0]) + 2 ** lv - 1) // 2 ** lv
        w = (int(image_hw[1]) + 2 ** lv - 1) // 2 ** lv
        cfg.H[i], cfg.W[i], cfg.stride[i] = h, w, params.strides[i]
        cfg.anchor_off[i] = off
        off += h * w * cfg.A
        base = L.generate_anchors_f64(params.sizes[i], params.ratios, params.scales)
        for a in range(cfg.A):
            for j in range(4):
                cfg.base[i][a][j] = float(base[a, j])
    for i in range(len(levels), L.RTN_MAX_GROUPS + 1):
        cfg.anchor_off[i] = off
    return cfg, off


class Engine:
    def __init__(self, backbone="resnet50", num_classes=1, num_anchors=9, dtype="bf16", device=0, anchor_params=None):
        if not torch.cuda.is_available():
            raise RuntimeError("the RetinaNet engine needs a ROCm GPU: no CPU fallback exists")
        if backbone.split("_")[0] not in Wt.STAGE_BLOCKS:
            raise ValueError("Backbone ('{}') not in allowed backbones ({}).".format(backbone, list(Wt.STAGE_BLOCKS)))
        self.backbone = backbone.split("_")[0]
        self.K, self.A = int(num_classes), int(num_anchors)
        self.dtype = dtype
        self.tdt, self.rdt = _TORCH_DT[dtype], _RTN_DT[dtype]
        self.device = torch.device("cuda", device)
        self.h = L.Handle(device)
        self.anchor_params = anchor_params or AnchorParams()
        self.w = {}
        self.plans = {}
        # A plan owns every activation buffer, descriptor and workspace of one (batch, canvas): several GB at the benchmark sizes.
        # csv_generator.compute_inputs pads each batch to ITS largest page, so a data set with varied page shapes shows a new canvas
        # nearly every step: the cache keeps the `max_plans` most recently used canvases and frees the rest (RTN_MAX_PLANS).
        self.max_plans = max(1, int(os.environ.get("RTN_MAX_PLANS", "4")))
        # detect() with `in_flight` > 1: consecutive batches run on `in_flight` buffer sets, each on ONE HIP stream of its own, so the
        # next batch's backbone fills the CUs the current batch's small-M layers and kernel tails leave idle (see detect()).
        self.in_flight = 1
        self.last_slot = 0
        self.trace_in_flight = None
        self._slots, self._next_slot = [], 0
        self.on_plan_evict = []        # weakref.WeakMethod callbacks key -> None (a Trainer drops its backward plan of the same canvas)
        self.state = None
        self.training = False          # set by trainer.Trainer: the pool then records its argmax taps
        self.two_streams = os.environ.get("RTN_TWO_STREAMS", "1") != "0"    # graph forks on side HIP streams (_schedule)
        self.fuse_stem = os.environ.get("RTN_FUSE_STEM", "1") != "0"        # inference/bf16: conv1+ReLU+pool1 in one kernel
        self.fuse_stem_2a = os.environ.get("RTN_FUSE_STEM_2A", "1") != "0"  # ... which also applies res2a_branch2a to the pooled pixels
        # training/bf16: the same kernel also records the pool's winning taps; conv1's output is never written and the pool's
        # backward takes its ReLU mask from pool1 (trainer.py)
        self.fuse_stem_train = os.environ.get("RTN_FUSE_STEM_TRAIN", "1") != "0"
        self.fuse_shortcut = os.environ.get("RTN_FUSE_SHORTCUT", "1") != "0"  # inference: branch1 folded into branch2c (dual-source GEMM)
        # inference/bf16: the identity blocks of the 64-channel stage as one kernel each (rtn_bottleneck64_fwd: branch2b -> branch2c
        # + shortcut -> the next block's branch2a)
        self.fuse_bottleneck = os.environ.get("RTN_FUSE_BOTTLENECK", "1") != "0"
        # ... and res2b_branch2a appended to res2a's fused block (its projection form with the three per-chunk filter sets streamed)
        self.fuse_proj_tail = os.environ.get("RTN_FUSE_PROJ_TAIL", "1") != "0"
        # bf16: an identity block's branch2c + Add + ReLU and the next block's branch2a as one launch in the 128-channel stage
        # (rtn_chain1x1_fwd; the 256-channel instance measured no faster and left the library: profiles/r4_seam_kernel.txt)
        self.fuse_chain = os.environ.get("RTN_FUSE_CHAIN", "1") != "0"
        self.weights_version = 0
        self.load_epoch = 0            # bumped by load_state(): a live Trainer re-derives its master copy / plans from it
        self._dual, self._dual_version = {}, -1
        self._side = None
        self.fp8_scales = None         # set by calibrate_fp8(): the head towers then run in fp8 (inference, bf16 engine)
        self._w8, self._w8_version = {}, -1
        self.fp8_backbone = False      # calibrate_fp8(..., backbone=True): the 3x3 branch2b layers with >= 128 channels too

    # ------------------------------------------------------------------ weights
    def load_state(self, state):
        """state: Keras-named dict (see weights.py). Packs every conv (BN folded) into ONE flat device buffer of
        forward weights (dtype of the path) and one flat f32 bias buffer; per-layer tensors are views into them, so an
        optimizer can rewrite all weights with a single kernel (trainer.py)."""
        if any(s["done"] is not None for s in self._slots):
            torch.cuda.synchronize(self.device)          # batches in flight read the buffers released below
        self._slots, self._next_slot = [], 0
        self.state = state
        self.w = {}
        self.layout = {}
        packed, woff, boff = [], 0, 0
        for (name, kh, kw, cin, cout, has_bias, bn) in Wt.conv_layers(self.backbone, self.K, self.A):
            bnp = None if bn is None else [state[bn + s] for s in ("/gamma", "/beta", "/moving_mean", "/moving_variance")]
            bias = state.get(name + "/bias") if has_bias else None
            pack = Wt.pack_stem if name == "conv1" else Wt.pack_conv
            wk, bk = pack(state[name + "/kernel"], bias, bnp, self.tdt, "cpu")
            packed.append((name, wk, bk, kh, kw, cin, cout))
            self.layout[name] = {"woff": woff, "rows": wk.shape[0], "K": wk.shape[1], "boff": boff, "has_bias": has_bias,
                                 "bn": bn, "kh": kh, "kw": kw, "cin": cin, "cout": cout}
            woff += wk.numel()
            boff += bk.numel()
        self.wflat = torch.empty(woff, dtype=self.tdt, device=self.device)
        self.bflat = torch.empty(boff, dtype=torch.float32, device=self.device)
        for (name, wk, bk, kh, kw, cin, cout) in packed:
            lo = self.layout[name]
            wv = self.wflat[lo["woff"]:lo["woff"] + wk.numel()].view(wk.shape)
            bv = self.bflat[lo["boff"]:lo["boff"] + bk.numel()]
            wv.copy_(wk)
            bv.copy_(bk)
            self.w[name] = (wv, bv, kh, kw, cin, cout)
        self.plans = {}
        self.weights_version += 1
        self.load_epoch += 1

    def _dual_weights(self):
        """K-concatenated filters of every stage's first block: [branch2c | branch1] along K, biases summed
        (rtn_conv1x1_dual_fwd).  Copies of the flat forward weights: refreshed whenever those change (load_state, an
        optimizer step)."""
        if self._dual_version != self.weights_version:
            for stage in range(4):
                s = str(stage + 2)
                bname = Wt.block_name(self.backbone, stage, 0)
                wc, bc = self.w["res%s%s_branch2c" % (s, bname)][:2]
                w1, b1 = self.w["res%s%s_branch1" % (s, bname)][:2]
                key = "res%s%s" % (s, bname)
                if key not in self._dual:
                    self._dual[key] = (torch.empty(wc.shape[0], wc.shape[1] + w1.shape[1], dtype=wc.dtype, device=wc.device),
                                       torch.empty_like(bc))
                wd, bd = self._dual[key]
                wd[:, :wc.shape[1]].copy_(wc)
                wd[:, wc.shape[1]:].copy_(w1)
                torch.add(bc, b1, out=bd)
            self._dual_version = self.weights_version
        return self._dual

    # ------------------------------------------------------------------ fp8 head towers (BASELINE.json configs[4])
    TOWERS = ("pyramid_regression", "pyramid_classification")

    def _fp8_on(self):
        return self.fp8_scales is not None and self.dtype == "bf16" and not self.training

    def _fp8_layers(self):
        names = ["%s_%d" % (prefix, i) for prefix in self.TOWERS for i in range(4)]
        if self.fp8_backbone:
            for stage, nblocks in enumerate(Wt.STAGE_BLOCKS[self.backbone]):
                if 64 * 2 ** stage >= 128:                    # the halo kernel needs whole 128-byte channel chunks
                    names += ["res%d%s_branch2b" % (stage + 2, Wt.block_name(self.backbone, stage, b)) for b in range(nblocks)]
            names.append("P3")
        return names

    def _fp8_weights(self):
        """e4m3 copies of the fp8 layers' (BN-folded) filters, one scale per tensor (448 / max |w|); rebuilt when the weights change."""
        if self._w8_version != self.weights_version or any(n not in self._w8 for n in self._fp8_layers()):
            for name in self._fp8_layers():
                wk = self.w[name][0].float()
                sw = 448.0 / max(float(wk.abs().max()), 1e-30)
                wq = torch.clamp(wk * sw, -448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8).contiguous()
                if name in self._w8 and self._w8[name][0].shape == wq.shape:
                    self._w8[name][0].copy_(wq)                        # descriptors of existing plans keep pointing at it
                    self._w8[name] = (self._w8[name][0], sw)
                else:
                    self._w8[name] = (wq, sw)
            self._w8_version = self.weights_version
        return self._w8

    def calibrate_fp8(self, images, margin=1.25, backbone=False):
        """Switch the two head towers (4 x [3x3 conv 256 + ReLU] each, model/defineModel.py:78-167 - 38 % of the network's FLOPs
        at 1024x1024 with ResNet-101) to fp8 e4m3: activations get ONE static scale per tensor, 448 / (margin x max |x|) over the
        five pyramid levels of a bf16 forward pass of `images`; weights one scale per layer.  The towers' first input is the
        quantised pyramid, their last layer writes bf16 for the (bf16) output convs.  backbone=True also runs every 3x3 branch2b
        layer with >= 128 channels in fp8: its producer branch2a writes e4m3 straight from its epilogue (rtn_conv2d_fwd_fp8out),
        branch2b writes bf16 for the 1x1 branch2c.  `images` may be a list of batches (scales from the maximum over all of
        them); calibrate_fp8(None) switches back."""
        self.fp8_scales = None
        self.fp8_backbone = bool(backbone) and images is not None
        self.plans = {k: v for k, v in self.plans.items() if not k[3]}     # bf16 plans (and a trainer's view of them) stay valid
        if images is None:
            return None
        if self.dtype != "bf16":
            raise ValueError("fp8 towers extend the bf16 engine")
        batches = list(images) if isinstance(images, (list, tuple)) else [images]     # several batches: the maximum over all
        amax = {}

        def see(key, tensors):
            amax[key] = max([amax.get(key, 0.0)] + [float(t.float().abs().max()) for t in tensors])

        for x in batches:
            B, H, W, _ = x.shape
            self.forward(x)
            torch.cuda.synchronize(self.device)
            plan = self._plan(B, H, W)
            see("in", plan["pyr"])
            for prefix, acts in plan["tower_acts"].items():
                for i, levels in enumerate(acts):
                    see((prefix, i), levels)
            if self.fp8_backbone:
                for name, a in plan["a_acts"].items():
                    see(("a", name), [a])
        scales = {k: 448.0 / (margin * max(v, 1e-20)) for k, v in amax.items()}
        self.fp8_scales = scales
        return scales

    def _conv8(self, name, groups, B, s_in, s_out, out_fp8, relu=True):
        """One fp8 layer (rtn_conv2d_fp8_fwd): bias (+ ReLU), fp8 or bf16 output."""
        wq, sw = self._fp8_weights()[name]
        _, bk, kh, kw, cin, cout = self.w[name]
        d = L.ConvDesc()
        for i, g in enumerate(groups):
            d.g[i] = g
        d.ngroups, d.batch, d.dtype = len(groups), B, L.RTN_FP8
        d.w, d.bias = wq.data_ptr(), bk.data_ptr()
        d.w_rows, d.N, d.KH, d.KW = wq.shape[0], cout, kh, kw
        d.Crun = d.pix_stride = cin
        d.sy = d.sx = 1
        d.pad_t, d.pad_l = (kh - 1) // 2, (kw - 1) // 2
        d.out_ld, d.flags = cout, (L.CONV_RELU if relu else 0)
        q = L.ConvFp8(acc_scale=1.0 / (s_in * sw), out_scale=s_out, out_dtype=L.RTN_FP8 if out_fp8 else L.RTN_BF16)
        meta = {"name": name, "xs": [g._x for g in groups], "ys": [g._out for g in groups], "res": [None] * len(groups),
                "kh": kh, "kw": kw, "cin": cin, "cout": cout, "B": B, "fp8": True, "s_in": s_in, "sw": sw}
        return ("conv8", d, name, meta, q)

    # ------------------------------------------------------------------ descriptors
    def _group(self, x, out, Hout, Wout, res=None, out_off=0, out_img_stride=None, res_hw=None):
        g = L.ConvGroup()
        B, Hin, Win, Cin = x.shape
        g.in_ = x.data_ptr(); g.in_elems = x.numel()
        g.out = out.data_ptr(); g.out_elems = out.numel()
        g.in_img_stride = Hin * Win * Cin
        g.in_row_stride = Win * Cin
        g.Hin, g.Win, g.Hout, g.Wout = Hin, Win, Hout, Wout
        g.out_img_stride = out.numel() // B if out_img_stride is None else out_img_stride
        g.out_off = out_off
        if res is not None:
            g.res = res.data_ptr(); g.res_elems = res.numel()
            g.res_img_stride = res.numel() // B
            g.res_ld = res.shape[3]
            g.Hres, g.Wres = (res.shape[1], res.shape[2]) if res_hw is None else res_hw
        g._x, g._out, g._res, g._off = x, out, res, out_off       # python-side refs for the backward graph
        return g

    def _conv(self, name, groups, B, stride=1, pad=(0, 0), flags=0, out_ld=None):
        """One fused conv launch; the torch tensors behind the groups are kept in the op's meta for the backward graph."""
        wk, bk, kh, kw, cin, cout = self.w[name]
        d = L.ConvDesc()
        for i, g in enumerate(groups):
            d.g[i] = g
        d.ngroups, d.batch, d.dtype = len(groups), B, self.rdt
        d.w, d.bias = wk.data_ptr(), bk.data_ptr()
        d.w_rows, d.N, d.KH, d.KW = wk.shape[0], cout, kh, kw
        d.Crun, d.pix_stride = cin, cin
        d.sy = d.sx = stride
        d.pad_t, d.pad_l = pad
        d.out_ld = cout if out_ld is None else out_ld
        d.flags = flags
        L.attach_conv_workspace(self.h, d)                 # split-K / tail-split slabs (P6, P7, res4-sized grids): caller-owned
        meta = {"name": name, "xs": [g._x for g in groups], "ys": [g._out for g in groups], "res": [g._res for g in groups],
                "offs": [g._off for g in groups], "stride": stride, "pad": pad, "flags": flags,
                "relu": bool(flags & L.CONV_RELU), "kh": kh, "kw": kw, "cin": cin, "cout": cout, "B": B,
                "out_ld": d.out_ld}
        return ("conv", d, name, meta)

    def _dual_op(self, fb, B):
        """First block of a stage with the projection shortcut folded into branch2c: y = relu([W2c | W1] . [b2 ; x(step)] + b)."""
        wd, bd = self._dual_weights()[fb["key"]]
        f = fb["f"]
        d = L.ConvDesc()
        d.g[0] = self._group(fb["b2"], fb["y"], fb["Ho"], fb["Wo"])
        d.ngroups, d.batch, d.dtype = 1, B, self.rdt
        d.w, d.bias = wd.data_ptr(), bd.data_ptr()
        d.w_rows, d.N, d.KH, d.KW = wd.shape[0], 4 * f, 1, 1
        d.Crun = d.pix_stride = f
        d.sy = d.sx = 1
        d.pad_t = d.pad_l = 0
        d.out_ld, d.flags = 4 * f, L.CONV_RELU
        x = fb["x"]
        s2 = L.ConvSrc2()
        s2.in_, s2.in_elems = x.data_ptr(), x.numel()
        s2.in_img_stride, s2.in_row_stride, s2.pix_stride = x.shape[1] * x.shape[2] * x.shape[3], x.shape[2] * x.shape[3], x.shape[3]
        s2.Hin, s2.Win, s2.C, s2.step = x.shape[1], x.shape[2], x.shape[3], fb["step"]
        L.attach_conv_workspace(self.h, d, s2)             # stream-K slabs of the small-M stages (res5a): caller-owned
        return ("dual", d, fb["key"] + "_branch2c+1", s2, {"xs": [fb["b2"], x], "ys": [fb["y"]]})

    def _bneck_proj_op(self, fb, B, keep_h1=False, nxt=None):
        """The FIRST block of the 64-channel stage as one launch: branch2b, then branch2c and the projection shortcut as one product
        over the K-concatenated filters of _dual_weights() (rtn_bottleneck64_fwd with p_in / wproj), and branch2a of the following
        identity block `nxt` when there is one."""
        wd, bd = self._dual_weights()[fb["key"]]
        w2b, b2b = self.w[fb["n2b"]][:2]
        a, x, y = fb["a"], fb["x"], fb["y"]
        d = L.BottleneckDesc()
        d.a_in, d.a_in_elems = a.data_ptr(), a.numel()
        d.p_in, d.p_in_elems = x.data_ptr(), x.numel()
        d.x_out, d.x_out_elems = y.data_ptr(), y.numel()
        d.w2b, d.b2b, d.w2c, d.b2c = w2b.data_ptr(), b2b.data_ptr(), wd.data_ptr(), bd.data_ptr()
        d.wproj, d.w2c_ld = wd.data_ptr() + 64 * wd.element_size(), wd.shape[1]
        d.batch, d.H, d.W, d.mid, d.dtype = B, fb["Ho"], fb["Wo"], 64, self.rdt
        outs = [y]
        if nxt is not None:
            w2a, b2a = self.w[nxt["n2a"]][:2]
            d.a_out, d.a_out_elems = nxt["a"].data_ptr(), nxt["a"].numel()
            d.w2a, d.b2a = w2a.data_ptr(), b2a.data_ptr()
            outs.append(nxt["a"])
        if keep_h1:                                      # training: branch2b's activation is an input of the backward pass
            d.h1_out, d.h1_out_elems = fb["b2"].data_ptr(), fb["b2"].numel()
            outs.append(fb["b2"])
        return ("bneck", d, fb["n2b"] + "+2c+1" + ("+next2a" if nxt is not None else ""),
                {"xs": [a, x], "ys": outs, "B": B, "H": fb["Ho"], "W": fb["Wo"], "tail": nxt is not None, "proj": True, "h1": keep_h1})

    def _bneck_op(self, blk, nxt, B, keep_h1=False):
        """An identity block of the 64-channel stage as one launch (rtn_bottleneck64_fwd): branch2b + branch2c + Add + ReLU of
        `blk`, and branch2a of the following identity block `nxt` when there is one."""
        w2b, b2b = self.w[blk["n2b"]][:2]
        w2c, b2c = self.w[blk["n2c"]][:2]
        a, x, y = blk["a"], blk["x"], blk["y"]
        d = L.BottleneckDesc()
        d.a_in, d.a_in_elems = a.data_ptr(), a.numel()
        d.x_in, d.x_in_elems = x.data_ptr(), x.numel()
        d.x_out, d.x_out_elems = y.data_ptr(), y.numel()
        d.w2b, d.b2b, d.w2c, d.b2c = w2b.data_ptr(), b2b.data_ptr(), w2c.data_ptr(), b2c.data_ptr()
        outs = [y]
        if nxt is not None:
            w2a, b2a = self.w[nxt["n2a"]][:2]
            d.a_out, d.a_out_elems = nxt["a"].data_ptr(), nxt["a"].numel()
            d.w2a, d.b2a = w2a.data_ptr(), b2a.data_ptr()
            outs.append(nxt["a"])
        d.batch, d.H, d.W, d.mid, d.dtype = B, blk["Ho"], blk["Wo"], 64, self.rdt
        if keep_h1:
            d.h1_out, d.h1_out_elems = blk["b2"].data_ptr(), blk["b2"].numel()
            outs.append(blk["b2"])
        name = blk["n2b"] + "+2c" + ("+next2a" if nxt is not None else "")
        return ("bneck", d, name, {"xs": [a, x], "ys": outs, "B": B, "H": blk["Ho"], "W": blk["Wo"], "tail": nxt is not None, "h1": keep_h1})

    def _chain_op(self, sm):
        """The seam between two identity blocks of the 128- / 256-channel stages as one launch (rtn_chain1x1_fwd): branch2c + Add +
        ReLU of the first, branch2a + ReLU of the second."""
        w2c, b2c = self.w[sm["n2c"]][:2]
        w2a, b2a = self.w[sm["n2a"]][:2]
        h, x, y, a = sm["b2"], sm["x"], sm["y"], sm["a"]
        d = L.ChainDesc()
        d.h_in, d.h_in_elems, d.x_in, d.x_in_elems = h.data_ptr(), h.numel(), x.data_ptr(), x.numel()
        d.x_out, d.x_out_elems, d.a_out, d.a_out_elems = y.data_ptr(), y.numel(), a.data_ptr(), a.numel()
        d.w2c, d.b2c, d.w2a, d.b2a = w2c.data_ptr(), b2c.data_ptr(), w2a.data_ptr(), b2a.data_ptr()
        d.pixels, d.mid, d.out, d.next, d.dtype = y.numel() // y.shape[3], sm["f"], 4 * sm["f"], sm["f"], self.rdt
        return ("chain", d, sm["n2c"] + "+next2a", {"xs": [h, x], "ys": [y, a], "pixels": int(d.pixels), "mid": sm["f"]})

    # ------------------------------------------------------------------ plan
    def _plan(self, B, H, W, slot=0):
        fp8_on = self._fp8_on()
        key = (B, H, W, fp8_on) if slot == 0 else (B, H, W, fp8_on, slot)      # slot: buffer set of a batch in flight (detect())
        if key in self.plans:
            self.plans[key] = self.plans.pop(key)          # most recently used last
            return self.plans[key]
        if self.state is None:
            raise RuntimeError("load_state() first")
        if len(self.plans) >= self.max_plans:
            # least recently used canvases go.  Their buffers may still be read by launches in flight on the side streams, and the
            # caching allocator would hand the memory to this plan's tensors on the current stream: drain the device first (a new
            # canvas costs a plan build anyway).
            torch.cuda.synchronize(self.device)
            while len(self.plans) >= self.max_plans:
                old = next(iter(self.plans))
                self.on_plan_evict = [r for r in self.on_plan_evict if r() is not None]     # weak references to bound methods
                for r in self.on_plan_evict:
                    r()(old)
                del self.plans[old]
        dev, tdt = self.device, self.tdt
        ops, keep = [], []

        def buf(*shape, dtype=None):
            t = torch.empty(*shape, dtype=dtype or tdt, device=dev)
            keep.append(t)
            return t

        # ---- stem: pack to [B][Hp][Wp][4], then 7x7/2 as an 8x1 implicit GEMM over 32-element runs
        H1, W1 = (H + 1) // 2, (W + 1) // 2
        Hp = max(H + 6, 2 * (H1 - 1) + 8)
        Wp = max(W + 6, 2 * (W1 - 1) + 8)
        Wp += Wp & 1
        xin = {"B": B, "H": H, "W": W, "Hp": Hp, "Wp": Wp}
        xp = torch.zeros(B * Hp * Wp * 4 + 64, dtype=tdt, device=dev)     # +64: the last pixel's 32-run stays in bounds
        keep.append(xp)
        c1 = buf(B, H1, W1, 64)
        wk, bk, *_ = self.w["conv1"]
        d = L.ConvDesc()
        g = L.ConvGroup()
        g.in_ = xp.data_ptr(); g.in_elems = xp.numel()
        g.out = c1.data_ptr(); g.out_elems = c1.numel()
        g.in_img_stride, g.in_row_stride = Hp * Wp * 4, Wp * 4
        g.Hin, g.Win, g.Hout, g.Wout = Hp, Wp, H1, W1
        g.out_img_stride = H1 * W1 * 64
        d.g[0] = g
        d.ngroups, d.batch, d.dtype = 1, B, self.rdt
        d.w, d.bias, d.w_rows, d.N = wk.data_ptr(), bk.data_ptr(), wk.shape[0], 64
        d.KH, d.KW, d.Crun, d.pix_stride = 8, 1, 32, 4
        d.sy = d.sx = 2
        d.pad_t = d.pad_l = 0
        d.out_ld, d.flags = 64, L.CONV_RELU
        ops.append(("pack", xp, xin))
        ops.append(("conv", d, "conv1", {"name": "conv1", "xs": [xp], "ys": [c1], "res": [None], "offs": [0], "stride": 2,
                                          "pad": (0, 0), "flags": L.CONV_RELU, "relu": True, "kh": 8, "kw": 1, "cin": 32,
                                          "cout": 64, "B": B, "out_ld": 64, "stem": True}))
        # ---- pool1
        H2, W2 = (H1 + 1) // 2, (W1 + 1) // 2
        x = buf(B, H2, W2, 64)
        pool_idx = torch.empty(B * H2 * W2 * 64, dtype=torch.uint8, device=dev)     # winning taps (training mode only)
        keep.append(pool_idx)
        ops.append(("pool", c1, x, (B, H1, W1, 64), pool_idx))
        # inference, bf16: the three stem ops above collapse into one kernel (rtn_stem_conv_pool)
        stem_fused = ("stem", x, (B, H, W), wk, bk, xp, (Hp, Wp), None, pool_idx)       # [7]: (a_out, w2a, b2a) with fuse_stem == 2
        n_stem_ops = len(ops)
        # ---- bottleneck stages
        feats = []
        first_blocks = []
        blocks64 = []                    # identity blocks of the 64-channel stage: candidates for the fused bottleneck kernel
        seams = []                       # identity block -> next block in the 128- / 256-channel stages: branch2c + next branch2a as one launch
        prev_blk = None
        a_acts = {}
        for stage, nblocks in enumerate(Wt.STAGE_BLOCKS[self.backbone]):
            f = 64 * 2 ** stage
            for block in range(nblocks):
                s, bname = str(stage + 2), Wt.block_name(self.backbone, stage, block)
                st = 2 if (block == 0 and stage > 0) else 1
                Hi, Wi = x.shape[1], x.shape[2]
                Ho, Wo = (Hi - 1) // st + 1, (Wi - 1) // st + 1
                n2a, n2b = "res%s%s_branch2a" % (s, bname), "res%s%s_branch2b" % (s, bname)
                if fp8_on and ("a", n2a) in self.fp8_scales:      # branch2a -> e4m3 -> fp8 branch2b -> bf16
                    a = buf(B, Ho, Wo, f, dtype=torch.uint8)
                    sa = self.fp8_scales[("a", n2a)]
                    ops.append(("convq",) + self._conv(n2a, [self._group(x, a, Ho, Wo)], B, stride=st, flags=L.CONV_RELU)[1:] + (sa,))
                    b2 = buf(B, Ho, Wo, f)
                    ops.append(self._conv8(n2b, [self._group(a, b2, Ho, Wo)], B, sa, 1.0, out_fp8=False))
                else:
                    a = buf(B, Ho, Wo, f)
                    ops.append(self._conv(n2a, [self._group(x, a, Ho, Wo)], B, stride=st, flags=L.CONV_RELU))
                    i_2a = len(ops) - 1
                    b2 = buf(B, Ho, Wo, f)
                    ops.append(self._conv(n2b, [self._group(a, b2, Ho, Wo)], B, pad=(1, 1), flags=L.CONV_RELU))
                    i_2b_cur = len(ops) - 1
                    if f >= 128:
                        a_acts[n2a] = a
                    if f == 64 and block > 0:
                        blocks64.append({"block": block, "i_2a": i_2a, "i_2b": len(ops) - 1, "a": a, "x": x, "b2": b2, "n2a": n2a, "n2b": n2b,
                                         "n2c": "res%s%s_branch2c" % (s, bname), "Ho": Ho, "Wo": Wo})
                if block == 0:
                    sc = buf(B, Ho, Wo, 4 * f)
                    ops.append(self._conv("res%s%s_branch1" % (s, bname), [self._group(x, sc, Ho, Wo)], B, stride=st))
                    i_b1 = len(ops) - 1
                else:
                    sc = x
                y = buf(B, Ho, Wo, 4 * f)
                ops.append(self._conv("res%s%s_branch2c" % (s, bname), [self._group(b2, y, Ho, Wo, res=sc)], B,
                                      flags=L.CONV_RELU | L.CONV_RES_SAME))
                if not (fp8_on and ("a", n2a) in self.fp8_scales):
                    cur_blk = {"stage": stage, "block": block, "i_2a": i_2a, "i_2c": len(ops) - 1, "a": a, "b2": b2, "x": sc, "y": y, "f": f,
                               "n2a": n2a, "n2c": "res%s%s_branch2c" % (s, bname)}
                    if prev_blk is not None and prev_blk["stage"] == stage and prev_blk["block"] >= 1 and prev_blk["block"] + 1 == block \
                            and L.lib.rtn_chain1x1_supported(f, 4 * f, f):
                        seams.append({"i_2c": prev_blk["i_2c"], "i_2a": i_2a, "b2": prev_blk["b2"], "x": prev_blk["x"], "y": prev_blk["y"],
                                      "a": a, "f": f, "n2c": prev_blk["n2c"], "n2a": n2a})
                    prev_blk = cur_blk
                if block == 0:
                    first_blocks.append({"key": "res%s%s" % (s, bname), "i_b1": i_b1, "i_2c": len(ops) - 1, "x": x, "b2": b2, "y": y,
                                         "Ho": Ho, "Wo": Wo, "step": st, "f": f, "a": a, "n2b": n2b, "n2a": n2a,
                                         "i_2a": None if (fp8_on and ("a", n2a) in self.fp8_scales) else i_2a,
                                         "i_2b": None if (fp8_on and ("a", n2a) in self.fp8_scales) else i_2b_cur})
                if blocks64 and blocks64[-1].get("y") is None and blocks64[-1]["n2c"] == "res%s%s_branch2c" % (s, bname):
                    blocks64[-1].update({"i_2c": len(ops) - 1, "y": y})
                x = y
            feats.append(x)
        C3, C4, C5 = feats[1], feats[2], feats[3]
        # ---- FPN (model/defineModel.py:183-203)
        def hw(t):
            return t.shape[1], t.shape[2]
        P5r = buf(B, *hw(C5), 256)
        ops.append(self._conv("C5_reduced", [self._group(C5, P5r, *hw(C5))], B))
        P5 = buf(B, *hw(C5), 256)
        ops.append(self._conv("P5", [self._group(P5r, P5, *hw(C5))], B, pad=(1, 1)))
        P4m = buf(B, *hw(C4), 256)
        ops.append(self._conv("C4_reduced", [self._group(C4, P4m, *hw(C4), res=P5r)], B, flags=L.CONV_RES_UPSAMPLE))
        P4 = buf(B, *hw(C4), 256)
        ops.append(self._conv("P4", [self._group(P4m, P4, *hw(C4))], B, pad=(1, 1)))
        p3_fp8 = fp8_on and ("a", "C3_reduced") in self.fp8_scales
        if p3_fp8:        # P3's input has no other reader and P3 itself only feeds the towers: C3_reduced -> e4m3 -> fp8 P3 -> e4m3
            s3 = self.fp8_scales[("a", "C3_reduced")]
            P3m = buf(B, *hw(C3), 256, dtype=torch.uint8)
            ops.append(("convq",) + self._conv("C3_reduced", [self._group(C3, P3m, *hw(C3), res=P4m)], B, flags=L.CONV_RES_UPSAMPLE)[1:] + (s3,))
            P3 = buf(B, *hw(C3), 256, dtype=torch.uint8)
            ops.append(self._conv8("P3", [self._group(P3m, P3, *hw(C3))], B, s3, self.fp8_scales["in"], out_fp8=True, relu=False))
        else:
            P3m = buf(B, *hw(C3), 256)
            ops.append(self._conv("C3_reduced", [self._group(C3, P3m, *hw(C3), res=P4m)], B, flags=L.CONV_RES_UPSAMPLE))
            P3 = buf(B, *hw(C3), 256)
            ops.append(self._conv("P3", [self._group(P3m, P3, *hw(C3))], B, pad=(1, 1)))
            a_acts["C3_reduced"] = P3m
        H6, W6 = (C5.shape[1] + 1) // 2, (C5.shape[2] + 1) // 2
        P6 = buf(B, H6, W6, 256)
        ops.append(self._conv("P6", [self._group(C5, P6, H6, W6)], B, stride=2,
                              pad=(same_pad_before(C5.shape[1], 3, 2), same_pad_before(C5.shape[2], 3, 2))))
        P6r = buf(B, H6, W6, 256)
        ops.append(("relu", P6, P6r))
        H7, W7 = (H6 + 1) // 2, (W6 + 1) // 2
        P7 = buf(B, H7, W7, 256)
        ops.append(self._conv("P7", [self._group(P6r, P7, H7, W7)], B, stride=2,
                              pad=(same_pad_before(H6, 3, 2), same_pad_before(W6, 3, 2))))
        pyr = [P3, P4, P5, P6, P7]
        # ---- anchors / outputs
        cfg, N = make_anchor_cfg((H, W), self.anchor_params)
        for i, p in enumerate(pyr):
            if (p.shape[1], p.shape[2]) != (cfg.H[i], cfg.W[i]):
                raise RuntimeError("pyramid level %d is %s, anchors expect %s" % (i + 3, hw(p), (cfg.H[i], cfg.W[i])))
        regression = buf(B, N, 4, dtype=torch.float32)
        classification = buf(B, N, self.K, dtype=torch.float32)
        # ---- heads: every layer is ONE grouped launch over the five levels (weights shared, :217)
        tower_ranges = []
        tower_acts = {}
        if fp8_on:                                       # the pyramid, quantised once for both towers
            pyr8 = [p if p.dtype == torch.uint8 else buf(B, *hw(p), 256, dtype=torch.uint8) for p in pyr]
            for p, p8 in zip(pyr, pyr8):
                if p8 is not p:
                    ops.append(("quant", p, p8, self.fp8_scales["in"]))
        for prefix, out_t, per_anchor, last_flags in (("pyramid_regression", regression, 4, L.CONV_OUT_F32),
                                                      ("pyramid_classification", classification, self.K,
                                                       L.CONV_OUT_F32 | L.CONV_SIGMOID)):
            tower_start = len(ops)
            cur = pyr8 if fp8_on else pyr
            tower_acts[prefix] = []
            for i in range(4):
                if fp8_on:
                    nxt = [buf(B, *hw(p), 256, dtype=torch.uint8 if i < 3 else tdt) for p in pyr]
                    groups = [self._group(ci, ni, *hw(ci)) for ci, ni in zip(cur, nxt)]
                    s_in = self.fp8_scales["in"] if i == 0 else self.fp8_scales[(prefix, i - 1)]
                    ops.append(self._conv8("%s_%d" % (prefix, i), groups, B, s_in, self.fp8_scales[(prefix, i)], out_fp8=i < 3))
                else:
                    nxt = [buf(B, *hw(p), 256) for p in pyr]
                    groups = [self._group(ci, ni, *hw(ci)) for ci, ni in zip(cur, nxt)]
                    ops.append(self._conv("%s_%d" % (prefix, i), groups, B, pad=(1, 1), flags=L.CONV_RELU))
                tower_acts[prefix].append(nxt)
                cur = nxt
            groups = [self._group(ci, out_t, *hw(ci), out_off=cfg.anchor_off[l] * per_anchor, out_img_stride=N * per_anchor)
                      for l, ci in enumerate(cur)]
            ops.append(self._conv(prefix, groups, B, pad=(1, 1), flags=last_flags, out_ld=self.A * per_anchor))
            tower_ranges.append((tower_start, len(ops)))
        ws_bytes = L.lib.rtn_detect_workspace_bytes(B, N, self.K)
        sched = self._schedule(ops)
        variants = {}
        bneck_ok = self.dtype == "bf16" and not fp8_on
        for fs in (0, 1, 2):                             # fuse_stem: 1 = conv1 + ReLU + pool1, 2 = ... + res2a_branch2a
            for fd in (False, True):                     # fuse_shortcut
                for fk in ((0, 1, 2) if bneck_ok else (0,)):      # fuse_bottleneck: 1 = inference, 2 = training (branch2b's output is kept)
                  for fc in ((0, 1) if (bneck_ok and seams) else (0,)):      # fuse_chain
                   for fp in ((0, 1) if (fk and fd) else (0,)):                    # fuse_proj_tail
                    if not fs and not fd and not fk and not fc:
                        continue
                    v = list(ops)
                    if fc:
                        for sm in seams:
                            v[sm["i_2c"]] = self._chain_op(sm)
                            v[sm["i_2a"]] = None
                    if fd:
                        for fb in first_blocks:
                            v[fb["i_2c"]] = self._dual_op(fb, B)
                            v[fb["i_b1"]] = None
                    if fk and fd:                            # res2a: branch2b + [branch2c | branch1] + ReLU as one launch
                        for fb in first_blocks:
                            if fb["f"] == 64 and fb["step"] == 1 and fb["i_2b"] is not None:
                                nx0 = blocks64[0] if (blocks64 and fp) else None      # res2b's branch2a rides along
                                v[fb["i_2b"]] = self._bneck_proj_op(fb, B, keep_h1=fk == 2, nxt=nx0)
                                v[fb["i_2c"]] = None
                                if nx0 is not None:
                                    v[nx0["i_2a"]] = None
                    if fk:
                        for bi, blk in enumerate(blocks64):
                            nxt = blocks64[bi + 1] if bi + 1 < len(blocks64) else None      # its branch2a rides along
                            v[blk["i_2b"]] = self._bneck_op(blk, nxt, B, keep_h1=fk == 2)
                            v[blk["i_2c"]] = None
                            if nxt is not None:
                                v[nxt["i_2a"]] = None
                    if fs:
                        sf = stem_fused
                        fb0 = first_blocks[0]
                        if fs == 2 and fb0["f"] == 64 and fb0["step"] == 1 and fb0["i_2a"] is not None and v[fb0["i_2a"]] is not None:
                            w2a, b2a = self.w[fb0["n2a"]][:2]             # the pooled pixels go through res2a_branch2a before they leave the registers
                            sf = stem_fused[:7] + ((fb0["a"], w2a, b2a), stem_fused[8])
                            v[fb0["i_2a"]] = None
                        v = [v[0], sf] + v[n_stem_ops:]                  # pack, then conv1 + ReLU + pool1 as one launch
                    v = [op for op in v if op is not None]
                    variants[(fs, fd, fk, fc, fp)] = (v, self._schedule(v))
        plan = {"ops": ops, "towers": tower_ranges, "tower_acts": tower_acts, "a_acts": a_acts, "fp8": fp8_on, "sched": sched, "keep": keep, "variants": variants, "xin": xin, "cfg": cfg, "N": N, "regression": regression,
                "classification": classification, "pyr": pyr, "feats": feats,
                "det_ws": torch.empty(ws_bytes, dtype=torch.uint8, device=dev), "det_ws_bytes": ws_bytes,
                "boxes": torch.empty(B, L.RTN_MAX_DET, 4, dtype=torch.float32, device=dev),
                "scores": torch.empty(B, L.RTN_MAX_DET, dtype=torch.float32, device=dev),
                "labels": torch.empty(B, L.RTN_MAX_DET, dtype=torch.int32, device=dev)}
        self.plans[key] = plan
        return plan

    # ------------------------------------------------------------------ lanes
    @staticmethod
    def _op_io(op):
        """(tensors read, tensors written) of a plan op, as data pointers."""
        kind = op[0]
        if kind in ("conv", "conv8", "convq"):
            m = op[3]
            reads = [t.data_ptr() for t in m["xs"]] + [t.data_ptr() for t in m["res"] if t is not None]
            return reads, [t.data_ptr() for t in m["ys"]]
        if kind == "quant":
            return [op[1].data_ptr()], [op[2].data_ptr()]
        if kind == "pack":
            return [], [op[1].data_ptr()]
        if kind == "stem":
            return [op[5].data_ptr()], [op[1].data_ptr(), op[8].data_ptr()] + ([op[7][0].data_ptr()] if op[7] is not None else [])
        if kind == "dual":
            return [t.data_ptr() for t in op[4]["xs"]], [t.data_ptr() for t in op[4]["ys"]]
        if kind in ("bneck", "chain"):
            return [t.data_ptr() for t in op[3]["xs"]], [t.data_ptr() for t in op[3]["ys"]]
        if kind == "pool":
            return [op[1].data_ptr()], [op[2].data_ptr(), op[4].data_ptr()]
        if kind == "relu":
            return [op[1].data_ptr()], [op[2].data_ptr()]
        raise RuntimeError("unknown op %r" % (kind,))

    @staticmethod
    def _lane_of(op):
        """HIP stream lane of an op.  The graph is a chain except where the reference's model forks
        (model/defineModel.py:183-249): the projection shortcut of a stage's first block (branch1, beside branch2a/2b),
        the P6 -> P7 and P5 / P4 branches of the pyramid (beside the top-down C4/C3 path) and the two head towers.  Those
        run on side lanes so that their launch latency and partly filled last rounds of workgroups overlap other work."""
        if op[0] == "relu":
            return 1                                              # between P6 and P7
        if op[0] not in ("conv", "conv8", "convq"):
            return 0
        name = op[2]
        if name.endswith("_branch1") or name in ("P6", "P7"):
            return 1
        if name in ("P5", "P4"):
            return 2
        if name.startswith("pyramid_classification"):
            return 1
        return 0

    def _schedule(self, ops):
        """Per op: (lane, [indices of ops on OTHER lanes whose completion it must wait for], needs_event).  Dependencies are
        read-after-write and write-after-write on the plan's buffers (every tensor of a plan is written by exactly one op
        of a forward pass, so there are no write-after-read hazards inside a pass; passes are joined on lane 0)."""
        lanes = [self._lane_of(op) for op in ops]
        writer = {}
        waits = []
        for i, op in enumerate(ops):
            reads, writes = self._op_io(op)
            deps = set()
            for ptr in reads + writes:
                j = writer.get(ptr)
                if j is not None and lanes[j] != lanes[i]:
                    deps.add(j)
            waits.append(sorted(deps))
            for ptr in writes:
                writer[ptr] = i
        # per (consumer lane, producer lane) only the latest producer matters: streams are in order
        slim = []
        for i, deps in enumerate(waits):
            latest = {}
            for j in deps:
                latest[lanes[j]] = max(latest.get(lanes[j], -1), j)
            slim.append(sorted(latest.values()))
        needs_event = set(j for deps in slim for j in deps)
        last_on_lane = {}
        for i, ln in enumerate(lanes):
            last_on_lane[ln] = i
        joins = sorted(i for ln, i in last_on_lane.items() if ln != 0)
        needs_event.update(joins)
        return {"lanes": lanes, "waits": slim, "events": needs_event, "joins": joins, "nlanes": max(lanes) + 1}

    # ------------------------------------------------------------------ run
    def _bind_stream(self):
        self.h.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def forward(self, images):
        """images: device tensor (B,H,W,3), float32 / bfloat16 (already normalised) or uint8 (raw 3-channel
        distance-transform page: the x/127.5-1 of model/utils.py:43-46 is fused into the stem packer).
        Returns device tensors regression (B,N,4) f32, classification (B,N,K) f32 — the training
        model's outputs in the reference's order (model/defineModel.py:244-249).
        The returned tensors are the plan's OWN output buffers for this (B,H,W): the next forward()/detect() call with the
        same shape overwrites them (no allocation per step).  Clone what must outlive the next call
        (Model.predict_on_batch copies to the host)."""
        if images.device.type != "cuda" or images.dim() != 4 or images.shape[3] != 3:
            raise ValueError("images must be a (B,H,W,3) tensor on the GPU")
        if images.dtype not in _SRC_DT:
            raise ValueError("unsupported image dtype %s" % images.dtype)
        images = images.contiguous()
        B, H, W, _ = images.shape
        if self._fp8_on() and self._w8_version != self.weights_version:
            self.plans = {k: v for k, v in self.plans.items() if not k[3]}      # new filters: new weight scales in the fp8 ops
        plan = self._plan(B, H, W)
        self._bind_stream()
        fused = self._fused()
        ops = plan["variants"][fused][0] if fused else plan["ops"]
        if fused and fused[1]:
            self._dual_weights()                         # refresh the concatenated filters if the weights changed
        if not self.two_streams:
            for op in ops:
                self._run_op(op, images)
            return plan["regression"], plan["classification"]
        sched = plan["variants"][fused][1] if fused else plan["sched"]
        main = torch.cuda.current_stream(self.device)
        if self._side is None:                               # three lanes that do not share a hardware queue with each other or with `main`
            self._side = self._concurrent_streams(3, beside=(main,))
        streams = [main] + self._side
        lanes, waits = sched["lanes"], sched["waits"]
        events = plan.setdefault(("events", fused), {i: torch.cuda.Event() for i in sched["events"]})
        fork = plan.setdefault("fork", torch.cuda.Event())
        fork.record(main)                                    # side lanes start after everything queued before this pass
        for st in streams[1:sched["nlanes"]]:
            st.wait_event(fork)
        bound = None
        for i, op in enumerate(ops):
            st = streams[lanes[i]]
            for j in waits[i]:
                st.wait_event(events[j])
            if bound is not st:
                self.h.set_stream(st.cuda_stream)
                bound = st
            self._run_op(op, images)
            if i in events:
                events[i].record(st)
        for j in sched["joins"]:
            main.wait_event(events[j])
        self.h.set_stream(main.cuda_stream)
        return plan["regression"], plan["classification"]

    def _fused(self):
        """(stem fused: 0 / 1 / 2 = with res2a_branch2a, shortcut fused, 64-channel bottleneck blocks fused) or None.  Training keeps
        every bottleneck tensor (the backward reads them) and, in fp32, conv1 / pool1 separate; the folded
        shortcut is used there too - no gradient needs the shortcut TENSOR, only its input and filters.  The fused stem and the fused
        bottleneck exist for bf16 only; the fp8 plan keeps its own branch2a / branch2b pairing."""
        stem16 = self.dtype == "bf16" and (not self.training or self.fuse_stem_train)
        fk = 0
        if self.fuse_bottleneck and self.dtype == "bf16" and not self._fp8_on():
            fk = 2 if self.training else 1               # training: the fused blocks also store branch2b's output for the backward pass
        fs = (2 if self.fuse_stem_2a else 1) if (self.fuse_stem and stem16) else 0
        # (training too: both tensors of a seam are written, which is all the backward pass reads)
        fc = 1 if (self.fuse_chain and self.dtype == "bf16" and not self._fp8_on()) else 0
        fp = 1 if (self.fuse_proj_tail and fk and self.fuse_shortcut) else 0
        key = (fs, self.fuse_shortcut, fk, fc, fp)
        return key if any(key) else None

    def active_ops(self, plan):
        """The op list forward() executes."""
        fused = self._fused()
        return plan["variants"][fused][0] if fused else plan["ops"]

    def _run_op(self, op, images):
        lib, h = L.lib, self.h
        kind = op[0]
        if kind == "conv":
            h.check(lib.rtn_conv2d_fwd(h.raw, C.byref(op[1])))
        elif kind == "dual":
            h.check(lib.rtn_conv1x1_dual_fwd(h.raw, C.byref(op[1]), C.byref(op[3])))
        elif kind == "bneck":
            h.check(lib.rtn_bottleneck64_fwd(h.raw, C.byref(op[1])))
        elif kind == "chain":
            h.check(lib.rtn_chain1x1_fwd(h.raw, C.byref(op[1])))
        elif kind == "conv8":
            h.check(lib.rtn_conv2d_fp8_fwd(h.raw, C.byref(op[1]), C.byref(op[4])))
        elif kind == "convq":
            h.check(lib.rtn_conv2d_fwd_fp8out(h.raw, C.byref(op[1]), op[4]))
        elif kind == "quant":
            h.check(lib.rtn_quantize_fp8(h.raw, op[1].data_ptr(), self.rdt, op[2].data_ptr(), op[1].numel(), op[3]))
        elif kind == "stem":
            Bn, Hn, Wn = op[2]
            if op[7] is not None or self.training:       # + res2a_branch2a on the pooled pixels; training: + the pool's winning taps
                a_out, w2a, b2a = op[7] if op[7] is not None else (None, None, None)
                h.check(lib.rtn_stem_conv_pool_branch2a(h.raw, op[5].data_ptr(), op[6][0], op[6][1], op[3].data_ptr(), op[3].shape[0],
                                                        op[4].data_ptr(), op[1].data_ptr(), Bn, Hn, Wn,
                                                        w2a.data_ptr() if w2a is not None else None, b2a.data_ptr() if b2a is not None else None,
                                                        a_out.data_ptr() if a_out is not None else None,
                                                        op[8].data_ptr() if self.training else None))
            else:
                h.check(lib.rtn_stem_conv_pool(h.raw, op[5].data_ptr(), op[6][0], op[6][1], op[3].data_ptr(), op[3].shape[0],
                                               op[4].data_ptr(), op[1].data_ptr(), Bn, Hn, Wn))
        elif kind == "pack":
            xi = op[2]
            h.check(lib.rtn_stem_pack(h.raw, images.data_ptr(), _SRC_DT[images.dtype], op[1].data_ptr(), self.rdt,
                                      xi["B"], xi["H"], xi["W"], xi["Hp"], xi["Wp"]))
        elif kind == "pool":
            Bn, Hi, Wi, Cc = op[3]
            if self.training:
                h.check(lib.rtn_maxpool3x3s2_tfsame_fwd_idx(h.raw, op[1].data_ptr(), op[2].data_ptr(), op[4].data_ptr(), self.rdt, Bn, Hi, Wi, Cc))
            else:
                h.check(lib.rtn_maxpool3x3s2_tfsame_fwd(h.raw, op[1].data_ptr(), op[2].data_ptr(), self.rdt, Bn, Hi, Wi, Cc))
        elif kind == "relu":
            h.check(lib.rtn_relu(h.raw, op[1].data_ptr(), op[2].data_ptr(), self.rdt, op[1].numel()))
        else:
            raise RuntimeError("unknown op %r" % (kind,))

    def profile_ops(self, images, reps=1):
        """Per-op device time: events recorded on the launch stream around every op of `reps` forward passes.
        Returns [(kind, total_ms_over_reps)] in execution order, plus ("detect", ms) for the post-processing."""
        images = images.contiguous()
        B, H, W, _ = images.shape
        plan = self._plan(B, H, W)
        self._bind_stream()
        ops = self.active_ops(plan)
        totals = [0.0] * (len(ops) + 1)
        for _ in range(reps):
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(ops) + 2)]
            evs[0].record()
            for i, op in enumerate(ops):
                self._run_op(op, images)
                evs[i + 1].record()
            self.postprocess(plan["cfg"], plan["regression"], plan["classification"], H, W, plan["boxes"], plan["scores"],
                             plan["labels"], plan["det_ws"])
            evs[-1].record()
            torch.cuda.synchronize()
            for i in range(len(ops) + 1):
                totals[i] += evs[i].elapsed_time(evs[i + 1])
        return [(op[0], totals[i]) for i, op in enumerate(ops)] + [("detect", totals[-1])]

    def detect(self, images, score_threshold=0.05, nms_threshold=0.5, max_detections=300):
        """Inference model outputs [boxes (B,300,4), scores (B,300), labels (B,300)] (model/defineModel.py:310-315).
        Views of the plan's output buffers: overwritten by the next call with the same (B,H,W) - clone to keep them.

        in_flight > 1 (throughput mode; default 1): the call enqueues this batch on the next of `in_flight` buffer sets and returns at
        once; batches of consecutive calls then overlap on the device (separate HIP streams, one per buffer set: stage-4/5 layers
        use 44-175 of 256 CUs and every kernel has a tail - the neighbouring batch's kernels run there; measured 4.07 -> 3.57 ms per
        batch of 8 at 800x1333, profiles/r4_in_flight.txt).  The page tensor is consumed in the caller's stream order (the stem packer
        runs on the caller's stream), so it may be overwritten as soon as the call returns.  The returned views belong to the buffer
        set: valid after join() (or any device synchronisation) and until the in_flight-th call after this one."""
        md = int(max_detections)
        if not 1 <= md <= L.RTN_MAX_DET:
            raise ValueError("max_detections must be in [1, %d]" % L.RTN_MAX_DET)
        if self.in_flight > 1 and not self.training:
            return self._detect_in_flight(images, score_threshold, nms_threshold, md)
        reg, cls = self.forward(images)
        B, H, W, _ = images.shape
        plan = self._plan(B, H, W)
        # the library writes (B, max_detections, .) densely: hand it views of that shape over the plan's buffers
        boxes = plan["boxes"].view(-1)[:B * md * 4].view(B, md, 4)
        scores = plan["scores"].view(-1)[:B * md].view(B, md)
        labels = plan["labels"].view(-1)[:B * md].view(B, md)
        self.postprocess(plan["cfg"], reg, cls, H, W, boxes, scores, labels, plan["det_ws"], score_threshold, nms_threshold, md)
        return boxes, scores, labels

    def _detect_in_flight(self, images, score_threshold, nms_threshold, md):
        if images.device.type != "cuda" or images.dim() != 4 or images.shape[3] != 3:
            raise ValueError("images must be a (B,H,W,3) tensor on the GPU")
        if images.dtype not in _SRC_DT:
            raise ValueError("unsupported image dtype %s" % images.dtype)
        images = images.contiguous()
        B, H, W, _ = images.shape
        if len(self._slots) != self.in_flight:
            self.join()
            self._slots = [{"stream": st_, "done": None, "fed": torch.cuda.Event()} for st_ in self._concurrent_streams(self.in_flight)]
            self._next_slot = 0
        si = self._next_slot
        self._next_slot = (si + 1) % self.in_flight
        self.last_slot = si                                  # wait_slot(last_slot): the host waits for THIS batch only
        slot = self._slots[si]
        if self._fp8_on() and self._w8_version != self.weights_version:
            self.join()
            self.plans = {k: v for k, v in self.plans.items() if not k[3]}
        plan = self._plan(B, H, W, slot=si + 1)             # slot 0 is forward()'s / the one-batch path's own buffer set
        fused = self._fused()
        ops = plan["variants"][fused][0] if fused else plan["ops"]
        caller = torch.cuda.current_stream(self.device)
        if fused and fused[1] and self._dual_version != self.weights_version:
            self.join()                                       # new weights: the concatenated filters are rewritten on the caller's
            self._dual_weights()                              # stream - nothing may be in flight on the old ones, and the batches
            caller.synchronize()                              # that follow on other streams must see the new ones
        if slot.get("consumed") is not None:
            # the packer below rewrites this buffer set's packed image: its previous batch must have read it (the stem, that batch's
            # first kernel on the slot's stream).  NOT the whole previous batch: that would chain the slots through the caller's
            # stream and keep the batches in lockstep (measured: no overlap at all with two in flight, profiles/r4_in_flight.txt)
            caller.wait_event(slot["consumed"])
        self.h.set_stream(caller.cuda_stream)
        first = 0
        if ops and ops[0][0] == "pack":                      # the only reader of `images`: in the caller's stream order
            self._run_op(ops[0], images)
            first = 1
        else:
            images = images.clone()                          # (paths without the packer: a private copy, made in the caller's order)
        slot["fed"].record(caller)
        st = slot["stream"]
        st.wait_event(slot["fed"])
        # (Staggering the batches by stage - a batch's backbone behind the previous batch's backbone, its towers behind the previous
        # towers - was measured too and is WORSE than letting the slots' streams drift: 4.27 against 3.55 ms with three in flight.)
        with torch.cuda.stream(st):
            self.h.set_stream(st.cuda_stream)
            if self.trace_in_flight is not None:             # tools/exp_in_flight_engine.py: (slot, start, end) timing events per batch
                t_ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                t_ev[0].record(st)
            for i in range(first, len(ops)):
                self._run_op(ops[i], images)
                if i == first:
                    slot["consumed"] = torch.cuda.Event()
                    slot["consumed"].record(st)
            if first == 0:
                images.record_stream(st)                     # the private copy is read on the slot's stream
            boxes = plan["boxes"].view(-1)[:B * md * 4].view(B, md, 4)
            scores = plan["scores"].view(-1)[:B * md].view(B, md)
            labels = plan["labels"].view(-1)[:B * md].view(B, md)
            self.postprocess(plan["cfg"], plan["regression"], plan["classification"], H, W, boxes, scores, labels, plan["det_ws"],
                             score_threshold, nms_threshold, md)
            slot["done"] = torch.cuda.Event()
            slot["done"].record(st)
            if self.trace_in_flight is not None:
                t_ev[1].record(st)
                self.trace_in_flight.append((si, t_ev[0], t_ev[1]))
        self.h.set_stream(caller.cuda_stream)
        return boxes, scores, labels

    def _concurrent_streams(self, n, beside=()):
        """n HIP streams whose kernels really run beside each other (and beside the streams in `beside`).  The runtime multiplexes HIP streams onto a few hardware queues
        (GPU_MAX_HW_QUEUES, 4 by default) and two streams that share one are served strictly in turn: batches "in flight" on such a
        pair do not overlap at all (measured: 4.28 against 3.60 ms per batch, and WHICH streams of torch's pool collide changes with
        the streams created before; profiles/r4_in_flight.txt).  So candidates are tested, once: a one-workgroup spin kernel
        (torch.cuda._sleep) on the candidate and on every stream already chosen must take the time of one, not of two."""
        sleep = getattr(torch.cuda, "_sleep", None)
        chosen = []
        if sleep is None or os.environ.get("RTN_STREAM_SELECT", "1") == "0":
            return [torch.cuda.Stream(device=self.device) for _ in range(n)]
        import time
        cycles = 400000

        def spin(streams):
            torch.cuda.synchronize(self.device)
            t0 = time.perf_counter()
            for st_ in streams:
                with torch.cuda.stream(st_):
                    sleep(cycles)
            for st_ in streams:
                st_.synchronize()
            return time.perf_counter() - t0
        for _ in range(8 * n):
            if len(chosen) == n:
                break
            cand = torch.cuda.Stream(device=self.device)
            spin([cand])                                             # first use of a stream: not timed
            one = min(spin([cand]) for _ in range(2))
            if all(min(spin([cand, c]) for _ in range(2)) < 1.5 * one for c in list(beside) + chosen):
                chosen.append(cand)
        while len(chosen) < n:                                       # (no overlapping set found: the batches still run, one queue at a time)
            chosen.append(torch.cuda.Stream(device=self.device))
        return chosen

    def wait_slot(self, si):
        """The HOST waits until the batch most recently given to buffer set `si` is done (its outputs may then be copied out)."""
        if 0 <= si < len(self._slots) and self._slots[si]["done"] is not None:
            self._slots[si]["done"].synchronize()

    def join(self):
        """The caller's current stream waits for every batch detect() has in flight (in_flight > 1)."""
        cur = torch.cuda.current_stream(self.device)
        for slot in self._slots:
            if slot["done"] is not None:
                cur.wait_event(slot["done"])

    def postprocess(self, cfg, regression, classification, H, W, boxes, scores, labels, ws, score_threshold=0.05,
                    nms_threshold=0.5, max_detections=300):
        self._bind_stream()
        B = regression.shape[0]
        self.h.check(L.lib.rtn_decode_filter_nms(self.h.raw, C.byref(cfg), B, classification.shape[2], regression.data_ptr(),
                                                  classification.data_ptr(), H, W, score_threshold, nms_threshold,
                                                  max_detections, boxes.data_ptr(), scores.data_ptr(), labels.data_ptr(),
                                                  ws.data_ptr(), ws.numel()))
