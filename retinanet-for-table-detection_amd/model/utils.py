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


def shift(shape, stride, anchors):
    """ model/utils.py:51-80 (the in-graph float32 twin of anchors.shift): anchors shifted over a (rows, cols) map,
    centres (i + 0.5) * stride, order y -> x -> anchor -> (shape[0] * shape[1] * A, 4) float32 (rtn_anchors_f32)."""
    import ctypes as C
    from . import anchors as _anchors
    base = np.asarray(anchors, np.float32)
    h = _rt.handle()
    cfg, n = _anchors._cfg([shape], [stride], [base.astype(np.float64)])
    out = torch.empty(n, 4, dtype=torch.float32, device="cuda")
    h.check(L.lib.rtn_anchors_f32(h.raw, C.byref(cfg), out.data_ptr()))
    return _rt.host(out)


def bbox_transform_inv(boxes, deltas, mean=None, std=None):
    """ model/utils.py:84-112: boxes (B, N, 4) + deltas * std + mean applied to the box width / height (rtn_regress_boxes,
    the kernel behind layers.RegressBoxes)."""
    import ctypes as C
    if mean is None:
        mean = [0, 0, 0, 0]
    if std is None:
        std = [0.2, 0.2, 0.2, 0.2]
    h = _rt.handle()
    a, r = _rt.dev(np.asarray(boxes), torch.float32), _rt.dev(np.asarray(deltas), torch.float32)
    if a.shape != r.shape or a.shape[-1] != 4:
        raise ValueError("boxes and deltas must both be (B, N, 4), got %s and %s" % (tuple(a.shape), tuple(r.shape)))
    out = torch.empty_like(a)
    m4 = (C.c_float * 4)(*[float(v) for v in mean])
    s4 = (C.c_float * 4)(*[float(v) for v in std])
    h.check(L.lib.rtn_regress_boxes(h.raw, a.data_ptr(), r.data_ptr(), a.numel() // 4, m4, s4, out.data_ptr()))
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


# ---- visualisation utilities (model/utils.py:267-373) and the per-page output step of RetinaNet.py:348-402 -----------------------
# Host-side raster work on the ORIGINAL page for a handful of boxes: NumPy + Pillow in place of cv2.
# Pixel parity with OpenCV's anti-aliased rectangle and Hershey text is not claimed (parity unpinned; SampleResults/*.png in the
# reference are qualitative); box geometry, colours, crop extents and output file names follow the reference.
def label_color(label):
    """ model/utils.py:267-283: colour `label` of 80 evenly spaced fully saturated hues, as a list of three ints."""
    import warnings
    if 0 <= label < 80:
        h = (label * (1.0 / 80)) * 6.0                     # np.arange(0, 1, 1/80)[label], then matplotlib's hsv_to_rgb at s = v = 1
        i, f = int(h) % 6, h - int(h)
        q, t = 1.0 - f, 1.0 - (1.0 - f)
        r, g, b = [(1, t, 0), (q, 1, 0), (0, 1, t), (0, q, 1), (t, 0, 1), (1, 0, q)][i]
        return [int(255 * r), int(255 * g), int(255 * b)]
    warnings.warn('Label {} has no color, returning default.'.format(label))
    return 0, 255, 0


def extract_box(image, box):
    """ model/utils.py:286-295: the sub-image image[y1:y2, x1:x2]."""
    b = np.array(box).astype(int)
    return image[b[1]:b[3], b[0]:b[2]]


def draw_box(image, box, color, thickness=5):
    """ model/utils.py:297-307.  The reference hands cv2.rectangle the colour 0 whatever `color` is (:307), so its boxes are black;
    kept.  The outline is centred on the box edges and `thickness` pixels wide, clipped to the image, drawn in place."""
    b = np.array(box).astype(int)
    H, W = image.shape[:2]
    lo, hi = thickness // 2, thickness - thickness // 2
    x1, y1, x2, y2 = min(b[0], b[2]), min(b[1], b[3]), max(b[0], b[2]), max(b[1], b[3])

    def band(ya, yb, xa, xb):
        ya, yb, xa, xb = max(ya, 0), min(yb, H), max(xa, 0), min(xb, W)
        if ya < yb and xa < xb:
            image[ya:yb, xa:xb] = 0

    band(y1 - lo, y1 + hi, x1 - lo, x2 + hi)
    band(y2 - lo, y2 + hi, x1 - lo, x2 + hi)
    band(y1 - lo, y2 + hi, x1 - lo, x1 + hi)
    band(y1 - lo, y2 + hi, x2 - lo, x2 + hi)


def draw_caption(image, box, caption, font_size=5, font_thickness=5):
    """ model/utils.py:310-318: `caption` above the box's top-left corner in (0, 0, 255), drawn in place (Pillow's built-in font
    scaled by font_size in place of FONT_HERSHEY_PLAIN)."""
    from PIL import Image, ImageDraw, ImageFont
    b = np.array(box).astype(int)
    font = ImageFont.load_default()
    probe = ImageDraw.Draw(Image.new("L", (1, 1)))
    l, t, r, btm = probe.textbbox((0, 0), caption, font=font)
    tile = Image.new("L", (max(r, 1), max(btm, 1)), 0)
    ImageDraw.Draw(tile).text((0, 0), caption, fill=255, font=font)
    s = max(1, int(round(font_size * 0.9)))
    mask = np.kron(np.asarray(tile) > 127, np.ones((s, s), bool))
    x0, y0 = int(b[0]), int(b[1]) - 10 - mask.shape[0]
    H, W = image.shape[:2]
    ys, xs = max(y0, 0), max(x0, 0)
    ye, xe = min(y0 + mask.shape[0], H), min(x0 + mask.shape[1], W)
    if ys < ye and xs < xe:
        m = mask[ys - y0:ye - y0, xs - x0:xe - x0]
        region = image[ys:ye, xs:xe]
        region[m] = (0, 0, 255) if image.ndim == 3 else 255


def draw_boxes(image, boxes, color, thickness=2):
    """ model/utils.py:321-330."""
    for b in boxes:
        draw_box(image, b, color, thickness=thickness)


def draw_detections(image, boxes, scores, labels, color=None, label_to_name=None, score_threshold=0.5):
    """ model/utils.py:333-352."""
    for i in np.where(scores > score_threshold)[0]:
        c = color if color is not None else label_color(labels[i])
        draw_box(image, boxes[i, :], color=c)
        caption = str(label_to_name(labels[i]) if label_to_name else labels[i]) + ': {0:.2f}'.format(scores[i])
        draw_caption(image, boxes[i, :], caption)


def draw_annotations(image, annotations, color=(0, 255, 0), label_to_name=None):
    """ model/utils.py:355-373."""
    if isinstance(annotations, np.ndarray):
        annotations = {'bboxes': annotations[:, :4], 'labels': annotations[:, 4]}
    assert('bboxes' in annotations)
    assert('labels' in annotations)
    assert(annotations['bboxes'].shape[0] == annotations['labels'].shape[0])
    for i in range(annotations['bboxes'].shape[0]):
        label = annotations['labels'][i]
        c = color if color is not None else label_color(label)
        draw_caption(image, annotations['bboxes'][i], '{}'.format(label_to_name(label) if label_to_name else label))
        draw_box(image, annotations['bboxes'][i], color=c)


def write_image(path, image):
    """cv2.imwrite of a uint8 (H,W[,3]) array whose channels the caller treats in OpenCV's B,G,R order."""
    from PIL import Image
    a = np.ascontiguousarray(image)
    Image.fromarray(a[:, :, ::-1] if a.ndim == 3 else a).save(path)


def render_detections(processed_page, draw, boxes, scores, labels, image_scale, result_dir, image_name, labels_to_names=None,
                      score_threshold=0.6):
    """The output step of test_image (RetinaNet.py:366-402) for one page.  boxes/scores/labels: the (1,300,·) detections of the
    page at network scale (score-descending).  Divides the boxes by image_scale, walks them until the first score below
    score_threshold, draws box + caption on `draw` (the original page, modified in place), and writes
    result_dir/detections_cropped/<head>_<k><tail> per table (cropped AFTER the box outline is drawn, as the reference does),
    <head>_noDete_minScore-_<score><tail> when the first detection already fails, and result_dir/detections_inImage/<image_name>.
    Returns the list of kept (box int[4], score, label)."""
    import os
    labels_to_names = labels_to_names or {0: 'table'}
    head, tail = os.path.splitext(image_name)
    os.makedirs(os.path.join(result_dir, "detections_cropped"), exist_ok=True)
    os.makedirs(os.path.join(result_dir, "detections_inImage"), exist_ok=True)
    boxes = np.asarray(boxes, np.float32) / np.float32(image_scale)
    kept, k = [], 0
    for box, score, label in zip(boxes[0], np.asarray(scores)[0], np.asarray(labels)[0]):
        if score < score_threshold:
            if k == 0:
                write_image(os.path.join(result_dir, "detections_cropped", "{}_{}_{}{}".format(head, "noDete_minScore-", score, tail)), draw)
            break
        b = box.astype(int)
        draw_box(draw, b, color=label_color(int(label)))
        write_image(os.path.join(result_dir, "detections_cropped", "{}_{}{}".format(head, k, tail)), extract_box(draw, b))
        draw_caption(draw, b, "{} {:.3f}".format(labels_to_names[int(label)], score))
        kept.append((b, float(score), int(label)))
        k += 1
    write_image(os.path.join(result_dir, 'detections_inImage', image_name), draw)
    return kept
