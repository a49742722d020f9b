"""model/anchors.py of the reference on the device (same names / arguments / assertions; model/anchors.py:7-313)."""
import ctypes as C

import numpy as np
import torch

from . import _rt
from . import utils

L = _rt.L


class AnchorParameters:
    """ The parameteres that define how anchors are generated (model/anchors.py:7-22)."""

    def __init__(self, sizes, strides, ratios, scales):
        self.sizes = sizes
        self.strides = strides
        self.ratios = ratios
        self.scales = scales

    def num_anchors(self):
        return len(self.ratios) * len(self.scales)


AnchorParameters_default = AnchorParameters(
    sizes=[32, 64, 128, 256, 512],
    strides=[8, 16, 32, 64, 128],
    ratios=np.array([0.5, 1, 2], 'float32'),
    scales=np.array([2 ** 0, 2 ** (1.0 / 3.0), 2 ** (2.0 / 3.0)], 'float32'),
)


def _gt_device(annotations_group, image_group):
    B = len(image_group)
    gb = np.zeros((B, L.RTN_MAX_GT, 4), np.float64)
    gl = np.zeros((B, L.RTN_MAX_GT), np.int32)
    gc = np.zeros((B,), np.int32)
    hw = np.zeros((B, 2), np.int32)
    for i, (image, ann) in enumerate(zip(image_group, annotations_group)):
        n = ann['bboxes'].shape[0]
        if n > L.RTN_MAX_GT:
            raise ValueError("at most %d ground-truth boxes per image are supported, got %d" % (L.RTN_MAX_GT, n))
        if n:
            gb[i, :n] = ann['bboxes']
            gl[i, :n] = np.asarray(ann['labels']).astype(int)
        gc[i] = n
        shp = image.shape
        hw[i] = (shp[0], shp[1]) if shp else (2 ** 30, 2 ** 30)
    return [_rt.dev(a, t) for a, t in ((gb, torch.float64), (gl, torch.int32), (gc, torch.int32), (hw, torch.int32))]


def anchor_targets_bbox(anchors, image_group, annotations_group, num_classes, negative_overlap=0.4, positive_overlap=0.5):
    """ Generate anchor targets for bbox detection (model/anchors.py:36-92). Returns (regression_batch, labels_batch)."""
    assert(len(image_group) == len(annotations_group)), "The length of the images and annotations need to be equal."
    assert(len(annotations_group) > 0), "No data received to compute anchor targets for."
    for annotations in annotations_group:
        assert('bboxes' in annotations), "Annotations should contain bboxes."
        assert('labels' in annotations), "Annotations should contain labels."
    h = _rt.handle()
    B, N = len(image_group), anchors.shape[0]
    ad = _rt.dev(anchors, torch.float64)
    gb, gl, gc, hw = _gt_device(annotations_group, image_group)
    reg = torch.empty(B, N, 5, dtype=torch.float32, device="cuda")
    lab = torch.empty(B, N, num_classes + 1, dtype=torch.float32, device="cuda")
    h.check(L.lib.rtn_anchor_targets_explicit(h.raw, ad.data_ptr(), N, B, num_classes, gb.data_ptr(), gl.data_ptr(), gc.data_ptr(),
                                              hw.data_ptr(), negative_overlap, positive_overlap, reg.data_ptr(), lab.data_ptr()))
    return _rt.host(reg), _rt.host(lab)


def compute_gt_annotations(anchors, annotations, negative_overlap=0.4, positive_overlap=0.5):
    """ model/anchors.py:96-117 -> (positive_indices, ignore_indices, argmax_overlaps_inds)."""
    h = _rt.handle()
    N, G = anchors.shape[0], annotations.shape[0]
    ov = utils._compute_overlap_device(anchors, annotations)
    pos = torch.empty(N, dtype=torch.uint8, device="cuda")
    ign = torch.empty(N, dtype=torch.uint8, device="cuda")
    arg = torch.empty(N, dtype=torch.int64, device="cuda")
    h.check(L.lib.rtn_gt_annotations(h.raw, ov.data_ptr(), N, G, negative_overlap, positive_overlap, pos.data_ptr(), ign.data_ptr(), arg.data_ptr()))
    return _rt.host(pos).astype(bool), _rt.host(ign).astype(bool), _rt.host(arg)


def layer_shapes(image_shape, model):
    """ model/anchors.py:120-140: {layer name: output shape}; only the pyramid levels are meaningful here."""
    shapes = guess_shapes(image_shape, [3, 4, 5, 6, 7])
    out = {model.layers[0].name: (None,) + tuple(image_shape)}
    for lv, s in zip([3, 4, 5, 6, 7], shapes):
        out["P%d" % lv] = (None, int(s[0]), int(s[1]), 256)
    return out


def make_shapes_callback(model):
    """ model/anchors.py:143-151."""
    def get_shapes(image_shape, pyramid_levels):
        shape = layer_shapes(image_shape, model)
        return [shape["P{}".format(level)][1:3] for level in pyramid_levels]
    return get_shapes


def guess_shapes(image_shape, pyramid_levels):
    """ model/anchors.py:155-165."""
    image_shape = np.array(image_shape[:2])
    return [(image_shape + 2 ** x - 1) // (2 ** x) for x in pyramid_levels]


def _cfg(shapes, strides, bases):
    cfg = L.AnchorCfg()
    cfg.nlevels, cfg.A = len(shapes), bases[0].shape[0]
    off = 0
    for i, (shp, st, base) in enumerate(zip(shapes, strides, bases)):
        cfg.H[i], cfg.W[i], cfg.stride[i] = int(shp[0]), int(shp[1]), int(st)
        cfg.anchor_off[i] = off
        off += int(shp[0]) * int(shp[1]) * cfg.A
        for a in range(cfg.A):
            for j in range(4):
                cfg.base[i][a][j] = float(base[a, j])
    for i in range(len(shapes), L.RTN_MAX_GROUPS + 1):
        cfg.anchor_off[i] = off
    return cfg, off


def anchors_for_shape(image_shape, pyramid_levels=None, anchor_params=None, shapes_callback=None):
    """ model/anchors.py:169-204 -> (N, 4) float64."""
    if pyramid_levels is None:
        pyramid_levels = [3, 4, 5, 6, 7]
    if anchor_params is None:
        anchor_params = AnchorParameters_default
    if shapes_callback is None:
        shapes_callback = guess_shapes
    shapes = shapes_callback(image_shape, pyramid_levels)
    bases = [generate_anchors(base_size=anchor_params.sizes[i], ratios=anchor_params.ratios, scales=anchor_params.scales)
             for i in range(len(pyramid_levels))]
    if len(pyramid_levels) > L.RTN_MAX_GROUPS or bases[0].shape[0] > 16:
        raise ValueError("at most %d pyramid levels and 16 anchors per cell are supported" % L.RTN_MAX_GROUPS)
    h = _rt.handle()
    cfg, n = _cfg(shapes, anchor_params.strides[:len(pyramid_levels)], bases)
    out = torch.empty(n, 4, dtype=torch.float64, device="cuda")
    h.check(L.lib.rtn_anchors_f64(h.raw, C.byref(cfg), out.data_ptr()))
    return _rt.host(out)


def shift(shape, stride, anchors):
    """ model/anchors.py:208-238."""
    h = _rt.handle()
    cfg, n = _cfg([shape], [stride], [np.asarray(anchors, np.float64)])
    out = torch.empty(n, 4, dtype=torch.float64, device="cuda")
    h.check(L.lib.rtn_anchors_f64(h.raw, C.byref(cfg), out.data_ptr()))
    return _rt.host(out)


def generate_anchors(base_size=16, ratios=None, scales=None):
    """ model/anchors.py:243-278 (host arithmetic in the C library, bit-identical to NumPy's)."""
    if ratios is None:
        ratios = AnchorParameters_default.ratios
    if scales is None:
        scales = AnchorParameters_default.scales
    return L.generate_anchors_f64(base_size, ratios, scales)


def bbox_transform(anchors, gt_boxes, mean=None, std=None):
    """Compute bounding-box regression targets for an image (model/anchors.py:282-313)."""
    if mean is None:
        mean = np.array([0, 0, 0, 0])
    if std is None:
        std = np.array([0.2, 0.2, 0.2, 0.2])
    if isinstance(mean, (list, tuple)):
        mean = np.array(mean)
    elif not isinstance(mean, np.ndarray):
        raise ValueError('Expected mean to be a np.ndarray, list or tuple. Received: {}'.format(type(mean)))
    if isinstance(std, (list, tuple)):
        std = np.array(std)
    elif not isinstance(std, np.ndarray):
        raise ValueError('Expected std to be a np.ndarray, list or tuple. Received: {}'.format(type(std)))
    h = _rt.handle()
    n = anchors.shape[0]
    ad, gd = _rt.dev(anchors, torch.float64), _rt.dev(gt_boxes, torch.float64)
    out = torch.empty(n, 4, dtype=torch.float64, device="cuda")
    m4 = (C.c_double * 4)(*[float(v) for v in mean])
    s4 = (C.c_double * 4)(*[float(v) for v in std])
    h.check(L.lib.rtn_bbox_transform(h.raw, ad.data_ptr(), gd.data_ptr(), n, m4, s4, out.data_ptr()))
    return _rt.host(out)
