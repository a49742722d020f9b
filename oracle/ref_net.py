"""oracle/ref_net.py — CPU restatement (PyTorch-CPU, fp32 or fp64) of the reference's TF graph.

TEST INFRASTRUCTURE ONLY (see oracle/ref_numpy.py header).  "Parity unpinned": the reference holds
no fixture for network numerics and Keras/TF/keras_resnet are not installed, so this restates
  * keras_resnet ResNet50/101/152(include_top=False, freeze_bn=True)  — third-party, un-vendored,
    unpinned by the reference (call sites model/defineModel.py:2-3,376-380); structure as published
    by keras-resnet 0.1/0.2: ZeroPadding2D(3) + 7x7/2 'conv1' (no bias) + frozen BN
    (eps 1e-5) + ReLU + MaxPool 3x3/2 'same'; bottleneck_2d blocks [3,4,6,3] / [3,4,23,3] / [3,8,36,3]
    with the stride on the FIRST 1x1 of block 0 of stages >= 1 and explicit pad-1 3x3 convs;
  * the FPN  model/defineModel.py:170-205, UpsampleLike model/layers.py:89-98 (legacy TF nearest);
  * the head submodels model/defineModel.py:78-167 and their concatenation :208-228.
TF semantics restated: padding='same' (pad_before = floor(pad_total/2)), resize_images(NEAREST,
align_corners=False): src = min(floor(dst * (in/out) [f32]), in-1).

Weights come in a Keras-style state dict: '<layer>/kernel' HWIO, '<layer>/bias',
'<bn>/gamma|beta|moving_mean|moving_variance'.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
STAGE_BLOCKS = {"resnet50": [3, 4, 6, 3], "resnet101": [3, 4, 23, 3], "resnet152": [3, 8, 36, 3]}
NUMERICAL = {"resnet50": [False] * 4, "resnet101": [False, True, True, False], "resnet152": [False, True, True, False]}


def block_char(backbone, stage, block):
    if block > 0 and NUMERICAL[backbone][stage]:
        return "b%d" % block
    return chr(ord("a") + block)


def same_pads(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def _bf16(x):
    return x.to(torch.bfloat16).to(x.dtype)


class RefNet:
    def __init__(self, state, backbone="resnet50", num_classes=1, num_anchors=9, dtype=torch.float32, emulate_bf16=False):
        self.backbone = backbone
        self.K = num_classes
        self.A = num_anchors
        self.dtype = dtype
        self.emu = emulate_bf16
        self.s = {k: torch.as_tensor(np.asarray(v)).to(dtype) for k, v in state.items()}

    # ---- primitives
    def _q(self, x):
        return _bf16(x) if self.emu else x

    def conv(self, x, name, stride=1, pad="same", bn=None, relu=False, res=None, keep_f32=False):
        w = self.s[name + "/kernel"]                       # HWIO
        b = self.s.get(name + "/bias")
        kh, kw = w.shape[0], w.shape[1]
        wt = w.permute(3, 2, 0, 1).contiguous()            # OIHW
        bias = b
        if bn is not None:                                 # frozen BN: y = g (x - m) / sqrt(v + eps) + beta
            g, beta = self.s[bn + "/gamma"], self.s[bn + "/beta"]
            m, v = self.s[bn + "/moving_mean"], self.s[bn + "/moving_variance"]
            scale = g / torch.sqrt(v + BN_EPS)
            shift = beta - m * scale
            if self.emu:                                   # the product folds the scale into the bf16 weights
                wt = wt * scale.view(-1, 1, 1, 1)
                bias = shift if bias is None else bias * scale + shift
                scale = None
        if self.emu:
            wt = _bf16(wt)
        if pad == "same":
            pt, pb = same_pads(x.shape[2], kh, stride)
            pl, pr = same_pads(x.shape[3], kw, stride)
        else:
            pt = pb = pl = pr = int(pad)
        x = F.pad(x, (pl, pr, pt, pb))
        y = F.conv2d(x, wt, None, stride=stride)
        if bn is not None and not self.emu:
            y = y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
            if bias is not None:
                raise ValueError("conv with both bias and BN is not part of the graph")
        elif bias is not None:
            y = y + bias.view(1, -1, 1, 1)
        if res is not None:
            y = y + res
        if relu:
            y = torch.relu(y)
        return y if keep_f32 else self._q(y)

    @staticmethod
    def maxpool_same(x):
        pt, pb = same_pads(x.shape[2], 3, 2)
        pl, pr = same_pads(x.shape[3], 3, 2)
        x = F.pad(x, (pl, pr, pt, pb), value=float("-inf"))
        return F.max_pool2d(x, 3, 2)

    @staticmethod
    def upsample_like(src, target):
        ih, iw, oh, ow = src.shape[2], src.shape[3], target.shape[2], target.shape[3]
        sh = np.float32(ih) / np.float32(oh)
        sw = np.float32(iw) / np.float32(ow)
        ys = np.minimum(np.floor(np.arange(oh, dtype=np.float32) * sh).astype(np.int64), ih - 1)
        xs = np.minimum(np.floor(np.arange(ow, dtype=np.float32) * sw).astype(np.int64), iw - 1)
        return src[:, :, torch.as_tensor(ys)][:, :, :, torch.as_tensor(xs)]

    # ---- graph
    def backbone_features(self, x):
        x = self.conv(x, "conv1", stride=2, pad=3, bn="bn_conv1", relu=True)
        x = self.maxpool_same(x)
        feats = []
        filters = 64
        for stage, nblocks in enumerate(STAGE_BLOCKS[self.backbone]):
            for block in range(nblocks):
                sc = str(stage + 2)
                bc = block_char(self.backbone, stage, block)
                stride = 2 if (block == 0 and stage > 0) else 1
                y = self.conv(x, "res%s%s_branch2a" % (sc, bc), stride=stride, pad=0, bn="bn%s%s_branch2a" % (sc, bc), relu=True)
                y = self.conv(y, "res%s%s_branch2b" % (sc, bc), stride=1, pad=1, bn="bn%s%s_branch2b" % (sc, bc), relu=True)
                if block == 0:
                    short = self.conv(x, "res%s%s_branch1" % (sc, bc), stride=stride, pad=0, bn="bn%s%s_branch1" % (sc, bc))
                else:
                    short = x
                x = self.conv(y, "res%s%s_branch2c" % (sc, bc), stride=1, pad=0, bn="bn%s%s_branch2c" % (sc, bc), relu=True, res=short)
            feats.append(x)
            filters *= 2
        return feats                                        # C2..C5

    def pyramid(self, C3, C4, C5):
        P5r = self.conv(C5, "C5_reduced")
        P5 = self.conv(P5r, "P5")
        P4m = self.conv(C4, "C4_reduced", res=self.upsample_like(P5r, C4))
        P4 = self.conv(P4m, "P4")
        P3m = self.conv(C3, "C3_reduced", res=self.upsample_like(P4m, C3))
        P3 = self.conv(P3m, "P3")
        P6 = self.conv(C5, "P6", stride=2)
        P7 = self.conv(torch.relu(P6), "P7", stride=2)
        return [P3, P4, P5, P6, P7]

    def head(self, f, prefix, n_out, sigmoid):
        y = f
        for i in range(4):
            y = self.conv(y, "%s_%d" % (prefix, i), relu=True)
        y = self.conv(y, prefix, keep_f32=True)
        y = y.permute(0, 2, 3, 1).reshape(y.shape[0], -1, n_out)
        return torch.sigmoid(y) if sigmoid else y

    def forward(self, images_nhwc):
        """images (B,H,W,3) -> regression (B,N,4), classification (B,N,K)."""
        x = torch.as_tensor(np.asarray(images_nhwc)).to(self.dtype).permute(0, 3, 1, 2).contiguous()
        x = self._q(x)
        _, C3, C4, C5 = self.backbone_features(x)
        feats = self.pyramid(C3, C4, C5)
        reg = torch.cat([self.head(f, "pyramid_regression", 4, False) for f in feats], dim=1)
        cls = torch.cat([self.head(f, "pyramid_classification", self.K, True) for f in feats], dim=1)
        return reg, cls


# ----------------------------------------------------------------------------------------------------------------------
# Training oracle: the same graph under torch autograd + the losses of model/losses.py in torch (float64), and the
# optimizer of RetinaNet.py:130 restated from keras.optimizers.Adam.get_updates with clipnorm (global norm, Keras 2.x).
def focal_torch(y_true, y_pred, alpha=0.25, gamma=2.0):
    """model/losses.py:22-44 (probabilities in, epsilon clip with the float32 constants, see ref_numpy.focal_loss)."""
    labels, state = y_true[..., :-1], y_true[..., -1]
    keep = (state != -1).unsqueeze(-1)
    one = labels == 1
    af = torch.where(one, torch.full_like(y_pred, alpha), torch.full_like(y_pred, 1 - alpha))
    fw = torch.where(one, 1 - y_pred, y_pred)
    lo, hi = float(np.float32(1e-7)), float(np.float32(1) - np.float32(1e-7))
    pc = torch.clamp(y_pred, lo, hi)
    bce = -(labels * torch.log(pc) + (1 - labels) * torch.log(1 - pc))
    total = torch.where(keep, af * fw ** gamma * bce, torch.zeros_like(bce)).sum()
    return total / max(1.0, float((state == 1).sum()))


def smooth_l1_torch(y_true, y_pred, sigma=3.0):
    """model/losses.py:58-90."""
    s2 = sigma ** 2
    pos = y_true[..., 4] == 1
    d = (y_pred - y_true[..., :4]).abs()
    terms = torch.where(d < 1.0 / s2, 0.5 * s2 * d ** 2, d - 0.5 / s2)
    return terms[pos].sum() / max(1.0, float(pos.sum()))


def train_step_oracle(state, images_nhwc, regression_batch, labels_batch, backbone="resnet50", num_classes=1,
                      dtype=torch.float64):
    """One forward/backward of total = smooth_l1 + focal (RetinaNet.py:125-131). Returns (loss parts, {kernel/bias name: grad})."""
    net = RefNet(state, backbone, num_classes, dtype=dtype)
    train = {k: v for k, v in net.s.items() if k.endswith("/kernel") or k.endswith("/bias")}
    for v in train.values():
        v.requires_grad_(True)
    reg, cls = net.forward(images_nhwc)
    yr = torch.as_tensor(np.asarray(regression_batch)).to(dtype)
    yl = torch.as_tensor(np.asarray(labels_batch)).to(dtype)
    l_reg, l_cls = smooth_l1_torch(yr, reg), focal_torch(yl, cls)
    (l_reg + l_cls).backward()
    return (float(l_reg.detach()), float(l_cls.detach())), {k: v.grad.detach() for k, v in train.items() if v.grad is not None}


def train_step_oracle_per_image(state, images_nhwc, regression_batch, labels_batch, backbone="resnet50", num_classes=1,
                                dtype=torch.float64, progress=None):
    """train_step_oracle for batches too large to hold under autograd at once (800x1333: ~7 GB per image in float64).  Same
    mathematics: both losses are sums over anchors divided by max(1, #positive anchors of the WHOLE batch) (model/losses.py:39-44,
    87-90), so the batch gradient is the sum of per-image gradients of (image's loss sum / batch normaliser)."""
    net = RefNet(state, backbone, num_classes, dtype=dtype)
    train = {k: v for k, v in net.s.items() if k.endswith("/kernel") or k.endswith("/bias")}
    for v in train.values():
        v.requires_grad_(True)
    yr = torch.as_tensor(np.asarray(regression_batch)).to(dtype)
    yl = torch.as_tensor(np.asarray(labels_batch)).to(dtype)
    n_reg = max(1.0, float((yr[..., 4] == 1).sum()))
    n_cls = max(1.0, float((yl[..., -1] == 1).sum()))
    l_reg = l_cls = 0.0
    for b in range(yr.shape[0]):
        reg, cls = net.forward(np.asarray(images_nhwc[b:b + 1]))
        yrb, ylb = yr[b:b + 1], yl[b:b + 1]
        # the per-image functions divide by max(1, own positives): multiply that back and divide by the batch count
        own_reg = max(1.0, float((yrb[..., 4] == 1).sum()))
        own_cls = max(1.0, float((ylb[..., -1] == 1).sum()))
        lr_b = smooth_l1_torch(yrb, reg) * (own_reg / n_reg)
        lc_b = focal_torch(ylb, cls) * (own_cls / n_cls)
        (lr_b + lc_b).backward()
        l_reg += float(lr_b.detach())
        l_cls += float(lc_b.detach())
        if progress is not None:
            progress(b)
    return (l_reg, l_cls), {k: v.grad.detach() for k, v in train.items() if v.grad is not None}


def adam_clipnorm_oracle(params, grads, m, v, step, lr=1e-4, b1=0.9, b2=0.999, eps=1e-7, clipnorm=0.001, global_clip=True):
    """keras.optimizers.Adam with clipnorm (RetinaNet.py:130).  global_clip=True: standalone Keras 2.x get_gradients — every gradient
    scaled by clipnorm/norm when the GLOBAL norm exceeds clipnorm; False: tf.keras / Keras >= 2.4 — tf.clip_by_norm per tensor."""
    norm = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads.values()))
    c = clipnorm / norm if (clipnorm and norm > clipnorm) else 1.0
    lr_t = lr * math.sqrt(1 - b2 ** step) / (1 - b1 ** step)
    out = {}
    for k, p in params.items():
        if not global_clip and k in grads:
            nk = math.sqrt(float((grads[k].double() ** 2).sum()))
            c = clipnorm / nk if (clipnorm and nk > clipnorm) else 1.0
        g = grads[k].double() * c if k in grads else torch.zeros_like(p, dtype=torch.float64)
        m[k] = b1 * m.get(k, 0) + (1 - b1) * g
        v[k] = b2 * v.get(k, 0) + (1 - b2) * g * g
        out[k] = p.double() - lr_t * m[k] / (torch.sqrt(v[k]) + eps)
    return out, norm
