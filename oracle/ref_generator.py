"""oracle/ref_generator.py — CPU restatement (NumPy) of the reference's data-generator path, SURVEY.md §8(f) rank 3.

TEST INFRASTRUCTURE ONLY: imported by tests/ and tools/bench_generator.py's cpu leg, never by the product path.

Follows csv_generator.py (read_annotations :16-52, group_images :159-171, filter_annotations :192-218, random_transform_group_entry
:248-265, preprocess_group_entry :289-304, compute_inputs :320-336, compute_targets :353-370) and model/transform.py (matrices
:58-234, adjust_transform_for_image :324-340, apply_transform :343-362).

Pinning: the matrices, transform_aabb, filter_annotations, group_images and load_annotations are checked against vectors the
reference's own code produced (tests/golden/ref_generator_golden.npz, written by oracle/gen_golden_generator.py).
PARITY UNPINNED: warp_affine_u8 restates cv2.warpAffine from OpenCV's published implementation (imgwarp.cpp: the double-precision
inversion, AB_BITS = 10 / INTER_BITS = 5 fixed-point coordinates, round-half-even products, the 15-bit bilinear table and
borderInterpolate) — opencv-python is not installed and the reference holds no before/after pair of an augmented page, so nothing
here can confirm it bit for bit; likewise image decoding (cv2.imread) is left to the caller.
"""
import csv
import os
import random

import numpy as np

from . import ref_numpy, ref_preprocess


# ---- model/transform.py: matrices (float64) ----------------------------------------------------------------------------------
def _mat(r0, r1):
    return np.array([r0, r1, [0, 0, 1]])


def random_transform(prng, min_rotation=0, max_rotation=0, min_translation=(0, 0), max_translation=(0, 0), min_shear=0, max_shear=0,
                     min_scaling=(1, 1), max_scaling=(1, 1), flip_x_chance=0, flip_y_chance=0):
    """One draw of model/transform.py:190-234; seven uniform draws in this order."""
    a = prng.uniform(min_rotation, max_rotation)
    t = prng.uniform(np.array(min_translation), np.array(max_translation))
    s = prng.uniform(min_shear, max_shear)
    z = prng.uniform(np.array(min_scaling), np.array(max_scaling))
    fx = prng.uniform(0, 1) < flip_x_chance
    fy = prng.uniform(0, 1) < flip_y_chance
    chain = [_mat([np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0]),
             _mat([1, 0, t[0]], [0, 1, t[1]]),
             _mat([1, -np.sin(s), 0], [0, np.cos(s), 0]),
             _mat([z[0], 0, 0], [0, z[1], 0]),
             _mat([1 - 2 * fx, 0, 0], [0, 1 - 2 * fy, 0])]
    return np.linalg.multi_dot(chain)


def adjust_for_image(transform, height, width, relative_translation=True):
    """model/transform.py:324-340 (on a copy)."""
    m = np.array(transform, dtype=np.float64)
    if relative_translation:
        m[0:2, 2] *= [width, height]
    cx, cy = 0.5 * width, 0.5 * height
    return np.linalg.multi_dot([_mat([1, 0, cx], [0, 1, cy]), m, _mat([1, 0, -cx], [0, 1, -cy])])


def transform_aabb(transform, box):
    x1, y1, x2, y2 = box
    p = transform.dot(np.array([[x1, x2, x1, x2], [y1, y2, y2, y1], [1, 1, 1, 1]], dtype=np.float64))
    return [p[0].min(), p[1].min(), p[0].max(), p[1].max()]


# ---- cv2.warpAffine (INTER_NEAREST / INTER_LINEAR) on uint8, restated ---------------------------------------------------------
def invert_affine(matrix):
    M = np.array(matrix, np.float64)[:2].copy()
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    a11, a22 = M[1, 1] * D, M[0, 0] * D
    M[0, 0], M[1, 1] = a11, a22
    M[0, 1] *= -D
    M[1, 0] *= -D
    b1 = -M[0, 0] * M[0, 2] - M[0, 1] * M[1, 2]
    b2 = -M[1, 0] * M[0, 2] - M[1, 1] * M[1, 2]
    M[0, 2], M[1, 2] = b1, b2
    return M


def _border(p, n, mode):
    """borderInterpolate; mode 0 constant (-1 = outside), 1 replicate, 2 reflect101, 3 wrap."""
    p = p.astype(np.int64).copy()
    if mode == 1:
        return np.clip(p, 0, n - 1)
    if mode == 2:
        if n == 1:
            return np.zeros_like(p)
        for _ in range(64):
            bad = (p < 0) | (p >= n)
            if not bad.any():
                break
            p = np.where(p < 0, -p, p)
            p = np.where(p >= n, 2 * n - 2 - p, p)
        return p
    if mode == 3:
        return np.mod(p, n)
    return np.where((p < 0) | (p >= n), -1, p)


def warp_affine_u8(img, matrix, interpolation=1, border_mode=1, cval=0):
    """cv2.warpAffine(img, matrix[:2], (W, H), flags=interpolation, borderMode, borderValue=cval) for uint8 (H,W,C)."""
    img = np.asarray(img)
    assert img.dtype == np.uint8
    squeeze = img.ndim == 2
    if squeeze:
        img = img[..., None]
    H, W, C = img.shape
    M = invert_affine(matrix)
    cv = np.zeros(C, np.int64)
    cv[0] = int(np.clip(np.rint(cval), 0, 255))                    # Scalar(cval): first channel only
    x = np.arange(W, dtype=np.float64)
    y = np.arange(H, dtype=np.float64)
    rd = 16 if interpolation else 512
    adelta = np.rint(M[0, 0] * x * 1024.0).astype(np.int64)
    bdelta = np.rint(M[1, 0] * x * 1024.0).astype(np.int64)
    X0 = np.rint((M[0, 1] * y + M[0, 2]) * 1024.0).astype(np.int64) + rd
    Y0 = np.rint((M[1, 1] * y + M[1, 2]) * 1024.0).astype(np.int64) + rd
    sh = 5 if interpolation else 10
    X = (X0[:, None] + adelta[None, :]) >> sh
    Y = (Y0[:, None] + bdelta[None, :]) >> sh
    if interpolation:
        sx, sy, fa, fb = X >> 5, Y >> 5, X & 31, Y & 31
    else:
        sx, sy = X, Y
    sx, sy = np.clip(sx, -32768, 32767), np.clip(sy, -32768, 32767)

    def tap(ix, iy):
        ok = (ix >= 0) & (iy >= 0)
        v = img[np.where(ok, iy, 0), np.where(ok, ix, 0)].astype(np.int64)
        return np.where(ok[..., None], v, cv[None, None, :])

    if not interpolation:
        out = tap(_border(sx, W, border_mode), _border(sy, H, border_mode))
    else:
        w00 = np.minimum(32 * (32 - fa) * (32 - fb), 32767)
        w01, w10, w11 = 32 * fa * (32 - fb), 32 * (32 - fa) * fb, 32 * fa * fb
        x0, x1 = _border(sx, W, border_mode), _border(sx + 1, W, border_mode)
        y0, y1 = _border(sy, H, border_mode), _border(sy + 1, H, border_mode)
        acc = (tap(x0, y0) * w00[..., None] + tap(x1, y0) * w01[..., None] + tap(x0, y1) * w10[..., None]
               + tap(x1, y1) * w11[..., None] + (1 << 14)) >> 15
        out = np.clip(acc, 0, 255)
        if border_mode == 0:
            gone = (sx >= W) | (sx + 1 < 0) | (sy >= H) | (sy + 1 < 0)
            out = np.where(gone[..., None], cv[None, None, :], out)
    out = out.astype(np.uint8)
    return out[..., 0] if squeeze else out


# ---- csv_generator.py ---------------------------------------------------------------------------------------------------------
def read_annotations(csv_path, image_dir, image_size):
    """_read_annotations (csv_generator.py:16-52): the first line is skipped, rows are image_id,xmin,ymin,xmax,ymax,label, grouped
    by image_id in SORTED order (pandas groupby), images absent from image_dir (or not *.png) are dropped.
    image_size(path) -> (height, width) stands for cv2.imread(path).shape.  Returns a list of records."""
    present = set(n for n in os.listdir(image_dir) if n.endswith('.png'))
    rows = {}
    with open(csv_path, newline='') as f:
        rd = csv.reader(f)
        next(rd, None)
        for r in rd:
            if not r:
                continue
            rows.setdefault(r[0], []).append(([float(v) for v in r[1:5]], r[5]))
    out = []
    for name in sorted(rows):
        if name not in present:
            continue
        path = os.path.join(image_dir, name)
        h, w = image_size(path)
        out.append({"name": name, "path": path, "height": h, "width": w, "boxes": np.array([b for b, _ in rows[name]], np.float64).reshape(-1, 4),
                    "names": [c for _, c in rows[name]]})
    return out


def group_images(n, batch_size, method, ratios=None):
    """csv_generator.py:159-171, drawing from the global `random` like the reference."""
    order = list(range(n))
    if method == 'random':
        random.shuffle(order)
    elif method == 'ratio':
        order.sort(key=lambda i: ratios[i])
    return [[order[x % n] for x in range(i, i + batch_size)] for i in range(0, n, batch_size)]


def filter_annotations(image_shape, boxes, labels):
    """csv_generator.py:192-218 for one image -> (boxes, labels) with the invalid rows removed."""
    b = np.asarray(boxes, np.float64).reshape(-1, 4)
    bad = (b[:, 2] <= b[:, 0]) | (b[:, 3] <= b[:, 1]) | (b[:, 0] < 0) | (b[:, 1] < 0) | (b[:, 2] > image_shape[1]) | (b[:, 3] > image_shape[0])
    return b[~bad], np.asarray(labels)[~bad]


def compute_input_output(pages, boxes_list, labels_list, num_classes, transforms=None, interpolation=1, border_mode=1, cval=0,
                         relative_translation=True, min_side=800, max_side=1333):
    """Generator.compute_input_output (csv_generator.py:373-398) for one group.  pages: list of uint8 (H,W,3); transforms: None or one
    3x3 matrix per image as the transform generator yields them (before adjust_transform_for_image).
    Returns inputs f32 (B,Hmax,Wmax,3), regression (B,N,5), labels (B,N,K+1), and the per-image boxes after all steps."""
    imgs, anns, labs = [], [], []
    for i, page in enumerate(pages):
        b, l = filter_annotations(page.shape, boxes_list[i], labels_list[i])
        if transforms is not None:
            t = adjust_for_image(transforms[i], page.shape[0], page.shape[1], relative_translation)
            page = warp_affine_u8(page, t, interpolation, border_mode, cval)
            b = np.array([transform_aabb(t, bb) for bb in b], np.float64).reshape(-1, 4)
        x = ref_numpy.preprocess_custom_tf(page)
        scale = ref_numpy.compute_resize_scale(page.shape, min_side, max_side)
        imgs.append(ref_preprocess.resize_cubic(x, scale))
        anns.append(b * scale)
        labs.append(l)
    Hm, Wm = max(i.shape[0] for i in imgs), max(i.shape[1] for i in imgs)
    batch = np.zeros((len(imgs), Hm, Wm, 3), np.float32)
    for i, im in enumerate(imgs):
        batch[i, :im.shape[0], :im.shape[1]] = im
    anchors = ref_numpy.anchors_for_shape((Hm, Wm, 3))
    reg, lab = ref_numpy.anchor_targets(anchors, [im.shape[:2] for im in imgs], anns, labs, num_classes)
    return batch, reg, lab, anns
