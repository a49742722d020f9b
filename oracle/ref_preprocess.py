"""oracle/ref_preprocess.py — CPU restatement (NumPy) of the page preprocessing the reference delegates to OpenCV.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED at bit level: opencv-python is not installed (and is unpinned by the reference);
the algorithms below restate OpenCV's published implementations of the calls made at DetectTablesUtils.py:251-261 and
model/utils.py:152 (SURVEY.md §8a notes).  The only artefact of the reference that pins them is the lossy JPEG pair
data/orig/sample_0717_023_orig.jpg -> data/processed/sample_0717_023.jpg, checked at PSNR level (tests/test_preprocess_oracle.py).

  cvtColor(BGR2GRAY)     8-bit fixed point: (1868 B + 9617 G + 4899 R + 8192) >> 14
  adaptiveThreshold(255, GAUSSIAN_C, BINARY, 11, 2): mean = GaussianBlur 11x11 (sigma 2.0, float, BORDER_REPLICATE) rounded to
                         uint8; dst = 255 if src - mean > -2 else 0
  distanceTransform      two-pass chamfer in 16.16 fixed point: DIST_L2 mask 5 -> a=1, b=1.4, c=2.1969; DIST_L1 / DIST_C are
                         forced to 3x3 (a=1,b=2 / a=1,b=1); result = int * 2^-16 as float32
  imwrite(float image)   saturate_cast<uchar>: round half to even, clamp to [0,255]
  resize(INTER_CUBIC)    a = -0.75, src = (dst + 0.5) / scale - 0.5, replicate border, float32 separable
"""
import math

import numpy as np

F32 = np.float32
DIST_SHIFT = 16
INF = np.int64((2 ** 31 - 1) >> 2)


def bgr2gray(img):
    b, g, r = [img[..., i].astype(np.int64) for i in range(3)]
    return ((1868 * b + 9617 * g + 4899 * r + 8192) >> 14).astype(np.uint8)


def gaussian_kernel(n=11, sigma=-1.0):
    if sigma <= 0:
        sigma = ((n - 1) * 0.5 - 1) * 0.3 + 0.8
    # cv::getGaussianKernel: taps exp(-x^2 / (2 sigma^2)) in double, summed in tap order, divided, stored as float.  libm's exp
    # and a sequential sum (not numpy's vectorised exp / pairwise sum) so that the taps do not depend on the host's SIMD level.
    k = [math.exp(-0.5 / (sigma * sigma) * (i - (n - 1) * 0.5) ** 2) for i in range(n)]
    total = 0.0
    for v in k:
        total += v
    return np.array([v / total for v in k], dtype=np.float64).astype(F32)


def adaptive_threshold_gaussian(gray, block=11, c=2):
    k = gaussian_kernel(block)
    r = block // 2
    src = gray.astype(F32)
    p = np.pad(src, ((0, 0), (r, r)), mode="edge")
    rows = np.zeros_like(src)
    for i in range(block):
        rows = rows + p[:, i:i + src.shape[1]] * k[i]
    p = np.pad(rows, ((r, r), (0, 0)), mode="edge")
    blur = np.zeros_like(src)
    for i in range(block):
        blur = blur + p[i:i + src.shape[0], :] * k[i]
    mean = np.clip(np.rint(blur), 0, 255).astype(np.int32)
    return np.where(gray.astype(np.int32) - mean > -int(np.ceil(c)), 255, 0).astype(np.uint8)


def _metric(dist):
    one = 1 << DIST_SHIFT
    if dist == "L2":      # 5x5 mask
        return one, int(round(1.4 * one)), int(round(2.1969 * one))
    if dist == "L1":
        return one, 2 * one, None
    if dist == "C":
        return one, one, None
    raise ValueError(dist)


def _shift(row, k):
    """row shifted so that out[x] = row[x + k], INF outside."""
    out = np.full_like(row, INF)
    if k == 0:
        out[:] = row
    elif k > 0:
        out[:-k] = row[k:]
    else:
        out[-k:] = row[:k]
    return out


def distance_transform(binary, dist):
    """Two raster passes (forward: top-left -> bottom-right, backward: the mirror), row-vectorised: the in-row recurrence
    D[x] = min(t[x], D[x-1] + a) is a prefix minimum of t[x] - a*x."""
    a, b, c = _metric(dist)
    H, W = binary.shape
    zero = binary == 0
    D = np.full((H + 4, W), INF, dtype=np.int64)
    ax = a * np.arange(W, dtype=np.int64)
    for y in range(H):
        p1, p2 = D[y + 1], D[y]                      # rows y-1, y-2 (offset 2)
        t = np.minimum.reduce([_shift(p1, -1) + b, p1 + a, _shift(p1, 1) + b])
        if c is not None:
            t = np.minimum.reduce([t, _shift(p2, -1) + c, _shift(p2, 1) + c, _shift(p1, -2) + c, _shift(p1, 2) + c])
        t = np.where(zero[y], 0, np.minimum(t, INF))
        D[y + 2] = np.minimum.accumulate(t - ax) + ax
    for y in range(H - 1, -1, -1):
        n1, n2 = D[y + 3], D[y + 4]
        t = np.minimum.reduce([D[y + 2], _shift(n1, 1) + b, n1 + a, _shift(n1, -1) + b])
        if c is not None:
            t = np.minimum.reduce([t, _shift(n2, 1) + c, _shift(n2, -1) + c, _shift(n1, 2) + c, _shift(n1, -2) + c])
        t = np.minimum(t, INF)
        D[y + 2] = (np.minimum.accumulate((t + ax)[::-1])[::-1]) - ax
    return (D[2:H + 2].astype(F32) * F32(1.0 / (1 << DIST_SHIFT))).astype(F32)


def distance_transform_literal(binary, dist):
    """The textbook two-pass raster scan, pixel by pixel (small images only): pins the row-vectorised version above."""
    a, b, c = _metric(dist)
    H, W = binary.shape
    D = np.full((H + 4, W + 4), int(INF), dtype=np.int64)
    fwd = [(-1, -1, b), (-1, 0, a), (-1, 1, b), (0, -1, a)]
    if c is not None:
        fwd += [(-2, -1, c), (-2, 1, c), (-1, -2, c), (-1, 2, c)]
    for y in range(H):
        for x in range(W):
            if binary[y, x] == 0:
                D[y + 2, x + 2] = 0
            else:
                D[y + 2, x + 2] = min(int(INF), min(D[y + 2 + dy, x + 2 + dx] + w for dy, dx, w in fwd))
    for y in range(H - 1, -1, -1):
        for x in range(W - 1, -1, -1):
            D[y + 2, x + 2] = min(D[y + 2, x + 2], min(D[y + 2 - dy, x + 2 - dx] + w for dy, dx, w in fwd))
    return (D[2:H + 2, 2:W + 2].astype(F32) * F32(1.0 / (1 << DIST_SHIFT))).astype(F32)


def to_u8(f):
    return np.clip(np.rint(f), 0, 255).astype(np.uint8)


def preprocess_page(bgr_or_gray):
    """DetectTablesUtils.py:251-261 -> uint8 (H, W, 3) in OpenCV's B,G,R order = (L2, L1, C)."""
    gray = bgr2gray(bgr_or_gray) if bgr_or_gray.ndim == 3 else bgr_or_gray
    binary = adaptive_threshold_gaussian(gray)
    return np.stack([to_u8(distance_transform(binary, m)) for m in ("L2", "L1", "C")], axis=-1), binary


def cubic_coeffs(x):
    A = F32(-0.75)
    x = x.astype(F32)
    c0 = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A
    c1 = ((A + 2) * x - (A + 3)) * x * x + 1
    c2 = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1
    c3 = F32(1) - c0 - c1 - c2
    return [c.astype(F32) for c in (c0, c1, c2, c3)]


def resize_cubic(img, scale):
    """cv2.resize(img, None, fx=scale, fy=scale, interpolation=INTER_CUBIC) on a float32 (H,W,C) image (model/utils.py:152)."""
    img = np.asarray(img, dtype=F32)
    H, W = img.shape[:2]
    Wo, Ho = int(round(W * scale)), int(round(H * scale))     # saturate_cast<int>(ssize * inv_scale)
    inv = 1.0 / scale

    def axis(n_out, n_in):
        f = (np.arange(n_out, dtype=np.float64) + 0.5) * inv - 0.5
        s = np.floor(f).astype(np.int64)
        return s, cubic_coeffs((f - s))

    sx, cx = axis(Wo, W)
    sy, cy = axis(Ho, H)
    rows = np.zeros((H, Wo) + img.shape[2:], dtype=F32)
    for k in range(4):
        idx = np.clip(sx - 1 + k, 0, W - 1)
        rows = rows + img[:, idx] * cx[k].reshape((1, -1) + (1,) * (img.ndim - 2))
    out = np.zeros((Ho, Wo) + img.shape[2:], dtype=F32)
    for k in range(4):
        idx = np.clip(sy - 1 + k, 0, H - 1)
        out = out + rows[idx] * cy[k].reshape((-1, 1) + (1,) * (img.ndim - 2))
    return out.astype(F32)
