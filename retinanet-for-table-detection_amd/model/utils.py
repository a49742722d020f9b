"""model/utils.py of the reference: image utilities (:19-211) and model utilities (:218-259) on the device."""
import sys

import numpy as np
import torch

from . import _rt
from .Parameters import image_scaling_factor, image_subtraction_factor

L = _rt.L


def preprocess_image(x, mode='tf'):
    """ model/utils.py:19-47: float32 cast, then 'tf' x/127.5-1 | 'caffe' BGR mean subtraction | 'custom_tf' x/127.5-1."""
    x = np.asarray(x)
    code = {'tf': 0, 'caffe': 1, 'custom_tf': 2}.get(mode)
    if code is None:
        return x.astype(np.float32)
    h = _rt.handle()
    src = _rt.dev(x, torch.uint8 if x.dtype == np.uint8 else torch.float32)
    out = torch.empty(src.shape, dtype=torch.float32, device="cuda")
    h.check(L.lib.rtn_preprocess_image(h.raw, src.data_ptr(), 2 if x.dtype == np.uint8 else 1, out.data_ptr(), src.numel(), code,
                                       float(image_scaling_factor), float(image_subtraction_factor)))
    return _rt.host(out)


def compute_resize_scale(image_shape, min_side=800, max_side=1333):
    """ model/utils.py:116-137."""
    (rows, cols, _) = image_shape
    smallest_side = min(rows, cols)
    scale = min_side / smallest_side
    largest_side = max(rows, cols)
    if largest_side * scale > max_side:
        scale = max_side / largest_side
    return scale


def resize_image(img, min_side=800, max_side=1333):
    """ model/utils.py:140-154: bicubic (cv2.INTER_CUBIC) resize by the computed scale. Returns (image, scale)."""
    from . import preprocess
    scale = compute_resize_scale(img.shape, min_side=min_side, max_side=max_side)
    return preprocess.resize_cubic(img, scale), scale


def _compute_overlap_device(boxes1, boxes2):
    h = _rt.handle()
    a, b = _rt.dev(np.asarray(boxes1, np.float64), torch.float64), _rt.dev(np.asarray(boxes2, np.float64), torch.float64)
    out = torch.empty(a.shape[0], b.shape[0], dtype=torch.float32, device="cuda")
    h.check(L.lib.rtn_compute_overlap(h.raw, a.data_ptr(), b.data_ptr(), a.shape[0], b.shape[0], out.data_ptr()))
    return out


def compute_overlap(boxes1, boxes2):
    """ model/utils.py:180-211 -> (len(boxes1), len(boxes2)) float32 IoU."""
    if len(boxes1) == 0 or len(boxes2) == 0:
        return np.zeros((len(boxes1), len(boxes2)), dtype=np.float32)
    return _rt.host(_compute_overlap_device(boxes1, boxes2))


def convert_model(model, nms=True, class_specific_filter=True, anchor_params=None):
    """ model/utils.py:218-231: training model -> inference model."""
    from .defineModel import retinanet_bbox
    return retinanet_bbox(model=model, nms=nms, class_specific_filter=class_specific_filter, anchor_params=anchor_params)


def assert_training_model(model):
    """ model/utils.py:234-238."""
    assert(all(output in model.output_names for output in ['regression', 'classification'])), \
        "Input is not a training model (no 'regression' and 'classification' outputs were found, outputs are: {}).".format(model.output_names)


def check_training_model(model):
    """ model/utils.py:241-248."""
    try:
        assert_training_model(model)
    except AssertionError as e:
        print(e, file=sys.stderr)
        sys.exit(1)


def freeze(model):
    """ model/utils.py:251-259."""
    for layer in model.layers:
        layer.trainable = False
    return model
