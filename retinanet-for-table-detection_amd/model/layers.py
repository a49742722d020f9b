"""model/layers.py of the reference: the custom Keras layers as callables on NumPy arrays, executed by librtn.so.
Constructor arguments and get_config() follow the reference (model/layers.py:7-369)."""
import ctypes as C

import numpy as np
import torch

from . import _rt
from . import anchors as utils_anchors

L = _rt.L


class _Layer:
    def __init__(self, name=None, **kwargs):
        self.name = name or self.__class__.__name__.lower()
        self.trainable = True

    def get_config(self):
        return {'name': self.name, 'trainable': self.trainable}

    def __call__(self, inputs, **kwargs):
        return self.call(inputs, **kwargs)


class Anchors(_Layer):
    """ Generates float32 anchors for the shape of a feature map (model/layers.py:7-75)."""

    def __init__(self, size, stride, ratios=None, scales=None, *args, **kwargs):
        self.size = size
        self.stride = stride
        self.ratios = ratios
        self.scales = scales
        if ratios is None:
            self.ratios = utils_anchors.AnchorParameters_default.ratios
        elif isinstance(ratios, list):
            self.ratios = np.array(ratios)
        if scales is None:
            self.scales = utils_anchors.AnchorParameters_default.scales
        elif isinstance(scales, list):
            self.scales = np.array(scales)
        self.num_anchors = len(self.ratios) * len(self.scales)
        self.anchors = utils_anchors.generate_anchors(base_size=size, ratios=self.ratios, scales=self.scales)
        super(Anchors, self).__init__(*args, **kwargs)

    def call(self, inputs, **kwargs):
        features = np.asarray(inputs)
        h = _rt.handle()
        cfg, n = utils_anchors._cfg([features.shape[1:3]], [self.stride], [self.anchors])
        out = torch.empty(n, 4, dtype=torch.float32, device="cuda")
        h.check(L.lib.rtn_anchors_f32(h.raw, C.byref(cfg), out.data_ptr()))
        return np.tile(_rt.host(out)[None], (features.shape[0], 1, 1))

    def compute_output_shape(self, input_shape):
        if None not in input_shape[1:]:
            return (input_shape[0], int(np.prod(input_shape[1:3])) * self.num_anchors, 4)
        return (input_shape[0], None, 4)

    def get_config(self):
        config = super(Anchors, self).get_config()
        config.update({'size': self.size, 'stride': self.stride, 'ratios': self.ratios.tolist(), 'scales': self.scales.tolist()})
        return config


class UpsampleLike(_Layer):
    """ Nearest-neighbour resize of `source` to the spatial shape of `target` (model/layers.py:78-104)."""

    def call(self, inputs, **kwargs):
        source, target = inputs
        source = np.asarray(source, np.float32)
        B, Hs, Ws, Cc = source.shape
        Hd, Wd = np.asarray(target).shape[1:3]
        if Cc % 4:
            raise ValueError("channel count must be a multiple of 4")
        h = _rt.handle()
        s = _rt.dev(source, torch.float32)
        out = torch.empty(B, Hd, Wd, Cc, dtype=torch.float32, device="cuda")
        h.check(L.lib.rtn_upsample_nearest(h.raw, s.data_ptr(), out.data_ptr(), L.RTN_F32, B, Hs, Ws, Hd, Wd, Cc))
        return _rt.host(out)

    def compute_output_shape(self, input_shape):
        return (input_shape[0][0],) + tuple(input_shape[1][1:3]) + (input_shape[0][-1],)


class RegressBoxes(_Layer):
    """ Applies regression values to boxes (model/layers.py:107-150)."""

    def __init__(self, mean=None, std=None, *args, **kwargs):
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
        self.mean = mean
        self.std = std
        super(RegressBoxes, self).__init__(*args, **kwargs)

    def call(self, inputs, **kwargs):
        anchors, regression = inputs
        h = _rt.handle()
        a, r = _rt.dev(anchors, torch.float32), _rt.dev(regression, torch.float32)
        out = torch.empty_like(a)
        m4 = (C.c_float * 4)(*[float(v) for v in self.mean])
        s4 = (C.c_float * 4)(*[float(v) for v in self.std])
        h.check(L.lib.rtn_regress_boxes(h.raw, a.data_ptr(), r.data_ptr(), a.numel() // 4, m4, s4, out.data_ptr()))
        return _rt.host(out)

    def compute_output_shape(self, input_shape):
        return input_shape[0]

    def get_config(self):
        config = super(RegressBoxes, self).get_config()
        config.update({'mean': self.mean.tolist(), 'std': self.std.tolist()})
        return config


class ClipBoxes(_Layer):
    """ Clips boxes to the shape of the image tensor (model/layers.py:153-174)."""

    def call(self, inputs, **kwargs):
        image, boxes = inputs
        shape = np.asarray(image).shape
        h = _rt.handle()
        b = _rt.dev(boxes, torch.float32)
        out = torch.empty_like(b)
        h.check(L.lib.rtn_clip_boxes(h.raw, b.data_ptr(), b.numel() // 4, float(shape[2]), float(shape[1]), out.data_ptr()))
        return _rt.host(out)

    def compute_output_shape(self, input_shape):
        return input_shape[1]


def _filter_batch(boxes, classification, score_threshold, max_detections, nms_threshold):
    h = _rt.handle()
    b, c = _rt.dev(boxes, torch.float32), _rt.dev(classification, torch.float32)
    B, N, K = c.shape
    wsb = L.lib.rtn_detect_workspace_bytes(B, N, K)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    ob = torch.empty(B, max_detections, 4, dtype=torch.float32, device="cuda")
    os_ = torch.empty(B, max_detections, dtype=torch.float32, device="cuda")
    ol = torch.empty(B, max_detections, dtype=torch.int32, device="cuda")
    h.check(L.lib.rtn_filter_detections(h.raw, B, N, K, b.data_ptr(), c.data_ptr(), score_threshold, nms_threshold, max_detections,
                                        ob.data_ptr(), os_.data_ptr(), ol.data_ptr(), ws.data_ptr(), wsb))
    return _rt.host(ob), _rt.host(os_), _rt.host(ol)


def filter_detections(boxes, classification, other=None, class_specific_filter=True, nms=True, score_threshold=0.05,
                      max_detections=300, nms_threshold=0.5):
    """ model/layers.py:177-264 for one image: (num_boxes,4), (num_boxes,num_classes) -> [boxes, scores, labels] padded with -1."""
    if other:
        raise NotImplementedError("`other` tensors are not carried through the device NMS")
    if not class_specific_filter or not nms:
        raise NotImplementedError("only the reference's configuration (class_specific_filter=True, nms=True) runs on the device")
    b, s, l = _filter_batch(np.asarray(boxes)[None], np.asarray(classification)[None], score_threshold, max_detections, nms_threshold)
    return [b[0], s[0], l[0]]


class FilterDetections(_Layer):
    """ Score threshold + NMS + top-k (model/layers.py:267-369)."""

    def __init__(self, nms=True, class_specific_filter=True, nms_threshold=0.5, score_threshold=0.05, max_detections=300,
                 parallel_iterations=32, **kwargs):
        self.nms = nms
        self.class_specific_filter = class_specific_filter
        self.nms_threshold = nms_threshold
        self.score_threshold = score_threshold
        self.max_detections = max_detections
        self.parallel_iterations = parallel_iterations
        super(FilterDetections, self).__init__(**kwargs)

    def call(self, inputs, **kwargs):
        boxes, classification = inputs[0], inputs[1]
        if len(inputs) > 2 or not self.nms or not self.class_specific_filter:
            raise NotImplementedError("only [boxes, classification] with nms=True, class_specific_filter=True runs on the device")
        return list(_filter_batch(boxes, classification, self.score_threshold, self.max_detections, self.nms_threshold))

    def compute_output_shape(self, input_shape):
        return [(input_shape[0][0], self.max_detections, 4), (input_shape[1][0], self.max_detections),
                (input_shape[1][0], self.max_detections)]

    def compute_mask(self, inputs, mask=None):
        return (len(inputs) + 1) * [None]

    def get_config(self):
        config = super(FilterDetections, self).get_config()
        config.update({'nms': self.nms, 'class_specific_filter': self.class_specific_filter, 'nms_threshold': self.nms_threshold,
                       'score_threshold': self.score_threshold, 'max_detections': self.max_detections,
                       'parallel_iterations': self.parallel_iterations})
        return config
