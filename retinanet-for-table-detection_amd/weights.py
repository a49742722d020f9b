"""Layer table, Keras-default initialisation and device packing of the RetinaNet weights.

Layer names and kernel layouts are the reference checkpoint's (Keras HDF5 keys):
'<layer>/kernel' HWIO, '<layer>/bias', '<bn>/gamma|beta|moving_mean|moving_variance'.
  backbone : keras_resnet ResNet50/101/152 (model/defineModel.py:376-380), names conv1, bn_conv1,
             res{stage}{block}_branch{2a,2b,2c,1}, bn{stage}{block}_branch*.
  FPN      : C5_reduced, P5, C4_reduced, P4, C3_reduced, P3, P6, P7 (model/defineModel.py:183-203).
  heads    : pyramid_regression_{0..3}, pyramid_regression, pyramid_classification_{0..3},
             pyramid_classification (model/defineModel.py:101-117,155-163).
"""
import math

import numpy as np
import torch

BN_EPS = 1e-5
STAGE_BLOCKS = {"resnet50": [3, 4, 6, 3], "resnet101": [3, 4, 23, 3], "resnet152": [3, 8, 36, 3]}
_NUMERICAL = {"resnet50": [False, False, False, False], "resnet101": [False, True, True, False],
              "resnet152": [False, True, True, False]}
W_ROW_ALIGN = 128


def block_name(backbone, stage, block):
    if block > 0 and _NUMERICAL[backbone][stage]:
        return "b%d" % block
    return chr(ord("a") + block)


def conv_layers(backbone="resnet50", num_classes=1, num_anchors=9, feature_size=256):
    """[(name, kh, kw, cin, cout, has_bias, bn_name_or_None)] in execution order."""
    out = [("conv1", 7, 7, 3, 64, False, "bn_conv1")]
    cin = 64
    for stage, nblocks in enumerate(STAGE_BLOCKS[backbone]):
        f = 64 * 2 ** stage
        for block in range(nblocks):
            s, b = str(stage + 2), block_name(backbone, stage, block)
            out.append(("res%s%s_branch2a" % (s, b), 1, 1, cin, f, False, "bn%s%s_branch2a" % (s, b)))
            out.append(("res%s%s_branch2b" % (s, b), 3, 3, f, f, False, "bn%s%s_branch2b" % (s, b)))
            out.append(("res%s%s_branch2c" % (s, b), 1, 1, f, 4 * f, False, "bn%s%s_branch2c" % (s, b)))
            if block == 0:
                out.append(("res%s%s_branch1" % (s, b), 1, 1, cin, 4 * f, False, "bn%s%s_branch1" % (s, b)))
            cin = 4 * f
    fs = feature_size
    out += [("C5_reduced", 1, 1, 2048, fs, True, None), ("P5", 3, 3, fs, fs, True, None),
            ("C4_reduced", 1, 1, 1024, fs, True, None), ("P4", 3, 3, fs, fs, True, None),
            ("C3_reduced", 1, 1, 512, fs, True, None), ("P3", 3, 3, fs, fs, True, None),
            ("P6", 3, 3, 2048, fs, True, None), ("P7", 3, 3, fs, fs, True, None)]
    for i in range(4):
        out.append(("pyramid_regression_%d" % i, 3, 3, fs, fs, True, None))
    out.append(("pyramid_regression", 3, 3, fs, num_anchors * 4, True, None))
    for i in range(4):
        out.append(("pyramid_classification_%d" % i, 3, 3, fs, fs, True, None))
    out.append(("pyramid_classification", 3, 3, fs, num_anchors * num_classes, True, None))
    return out


def init_state(backbone="resnet50", num_classes=1, num_anchors=9, seed=0, randomize_bn=False, cls_bias=None, tame=False):
    """Seeded random weights following the Keras initialisers the reference gets by default
    (SURVEY.md §8a notes): keras_resnet convs he_normal, BN gamma 1 / beta 0 / mean 0 / var 1;
    FPN convs glorot_uniform + zero bias (model/defineModel.py:183-203); head convs N(0, 0.01) +
    zero bias, classification output bias -log((1-p)/p), p = 0.01 (model/initializers.py:19-22).
    randomize_bn / cls_bias / tame are test knobs: exercise BN folding; put scores above the 0.05 threshold;
    `tame` makes the random network well conditioned like a trained one (small gamma on the last BN of every
    residual block so the residual stream does not double per block, He-scaled head kernels so the regression
    outputs are O(1) instead of O(0.01)) — noise comparisons against the oracle are only meaningful then."""
    g = torch.Generator().manual_seed(seed)
    st = {}
    for (name, kh, kw, cin, cout, has_bias, bn) in conv_layers(backbone, num_classes, num_anchors):
        fan_in, fan_out = kh * kw * cin, kh * kw * cout
        if bn is not None:
            w = torch.randn(kh, kw, cin, cout, generator=g) * math.sqrt(2.0 / fan_in)
        elif name.startswith("pyramid_"):
            std = 0.01
            if tame:
                # He scaling keeps the O(8) pyramid magnitude through the stack; the output layers bring it to O(1)
                std = (math.sqrt(1.0 / fan_in) / 8.0 if name in ("pyramid_regression", "pyramid_classification")
                       else math.sqrt(2.0 / fan_in))
            w = torch.randn(kh, kw, cin, cout, generator=g) * std
        else:
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            w = (torch.rand(kh, kw, cin, cout, generator=g) * 2 - 1) * lim
        st[name + "/kernel"] = w.numpy()
        if has_bias:
            b = torch.zeros(cout)
            if name == "pyramid_classification":
                b += (-math.log((1 - 0.01) / 0.01)) if cls_bias is None else cls_bias
            st[name + "/bias"] = b.numpy()
        if bn is not None:
            if randomize_bn:
                gam = 0.5 + torch.rand(cout, generator=g)
                if tame and bn.endswith("branch2c"):
                    gam = gam * 0.25
                st[bn + "/gamma"] = gam.numpy()
                st[bn + "/beta"] = (0.1 * torch.randn(cout, generator=g)).numpy()
                st[bn + "/moving_mean"] = (0.1 * torch.randn(cout, generator=g)).numpy()
                st[bn + "/moving_variance"] = (0.5 + torch.rand(cout, generator=g)).numpy()
            else:
                st[bn + "/gamma"] = np.ones(cout, np.float32)
                st[bn + "/beta"] = np.zeros(cout, np.float32)
                st[bn + "/moving_mean"] = np.zeros(cout, np.float32)
                st[bn + "/moving_variance"] = np.ones(cout, np.float32)
    return st


def fold_bn(kernel_hwio, bias, bn):
    """Frozen BN (freeze_bn=True): y = gamma (x - mean) / sqrt(var + 1e-5) + beta, folded into w/bias."""
    w = torch.as_tensor(np.asarray(kernel_hwio), dtype=torch.float32)
    cout = w.shape[3]
    b = torch.zeros(cout) if bias is None else torch.as_tensor(np.asarray(bias), dtype=torch.float32)
    if bn is not None:
        gamma, beta, mean, var = [torch.as_tensor(np.asarray(t), dtype=torch.float32) for t in bn]
        scale = gamma / torch.sqrt(var + BN_EPS)
        w = w * scale.view(1, 1, 1, -1)
        b = b * scale + (beta - mean * scale)
    return w, b


def pack_conv(kernel_hwio, bias, bn, torch_dtype, device):
    """HWIO -> [w_rows][KH*KW*Cin] (K contiguous, rows padded with zeros to a multiple of 128) + f32 bias."""
    w, b = fold_bn(kernel_hwio, bias, bn)
    kh, kw, cin, cout = w.shape
    rows = -(-cout // W_ROW_ALIGN) * W_ROW_ALIGN
    wk = torch.zeros(rows, kh * kw * cin, dtype=torch.float32)
    wk[:cout] = w.permute(3, 0, 1, 2).reshape(cout, -1)
    bk = torch.zeros(rows, dtype=torch.float32)
    bk[:cout] = b
    return wk.to(torch_dtype).contiguous().to(device), bk.to(device)


def pack_stem(kernel_hwio, bias, bn, torch_dtype, device):
    """7x7x3 stem -> [128][8 kernel rows][32-element run]: run index = kw*4 + c (kw < 7, c < 3), the rest zero.
    Matches the [B][Hp][Wp][4] input image produced by rtn_stem_pack."""
    w, b = fold_bn(kernel_hwio, bias, bn)
    kh, kw, cin, cout = w.shape
    assert (kh, kw, cin) == (7, 7, 3)
    rows = -(-cout // W_ROW_ALIGN) * W_ROW_ALIGN
    wk = torch.zeros(rows, 8, 8, 4, dtype=torch.float32)          # [n][kh][kw][c]
    wk[:cout, :7, :7, :3] = w.permute(3, 0, 1, 2)
    bk = torch.zeros(rows, dtype=torch.float32)
    bk[:cout] = b
    return wk.reshape(rows, 256).to(torch_dtype).contiguous().to(device), bk.to(device)
