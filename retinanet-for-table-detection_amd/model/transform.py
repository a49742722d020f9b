"""Augmentation geometry of the reference's model/transform.py behind the same names.

The 3x3 homogeneous matrices are host NumPy (float64, a handful of flops per image); the draws come from the caller's PRNG in the
reference's order (rotation, translation x/y, shear, scaling x/y, flip x, flip y - model/transform.py:190-234), so a seeded
RandomState yields the reference's matrices bit for bit (tests/golden/ref_generator_golden.npz holds matrices produced by the
reference module itself).  apply_transform - cv2.warpAffine in the reference (model/transform.py:343-362) - runs on the device
through rtn_warp_affine_u8.  The colour effects (VisualEffect, model/transform.py:397-511) are not mirrored: the reference never
applies them (the call is commented out at csv_generator.py:385).
"""
import numpy as np

DEFAULT_PRNG = np.random

_BORDER_CV = {'constant': 0, 'nearest': 1, 'reflect': 4, 'wrap': 3}            # cv2.BORDER_* values
_INTER_CV = {'nearest': 0, 'linear': 1, 'cubic': 2, 'area': 3, 'lanczos4': 4}  # cv2.INTER_* values
_BORDER_RTN = {0: 0, 1: 1, 4: 2, 3: 3}                                          # cv2 code -> rtn_warp_affine_u8 border_mode


def colvec(*args):
    return np.array([args]).T


def _homogeneous(top, bottom):
    return np.array([list(top), list(bottom), [0, 0, 1]])


def rotation(angle):
    c, s = np.cos(angle), np.sin(angle)
    return _homogeneous((c, -s, 0), (s, c, 0))


def translation(translation):
    return _homogeneous((1, 0, translation[0]), (0, 1, translation[1]))


def shear(angle):
    return _homogeneous((1, -np.sin(angle), 0), (0, np.cos(angle), 0))


def scaling(factor):
    return _homogeneous((factor[0], 0, 0), (0, factor[1], 0))


def _random_vector(min, max, prng=DEFAULT_PRNG):
    lo, hi = np.array(min), np.array(max)
    assert lo.shape == hi.shape
    assert len(lo.shape) == 1
    return prng.uniform(lo, hi)


def random_rotation(min, max, prng=DEFAULT_PRNG):
    return rotation(prng.uniform(min, max))


def random_translation(min, max, prng=DEFAULT_PRNG):
    return translation(_random_vector(min, max, prng))


def random_shear(min, max, prng=DEFAULT_PRNG):
    return shear(prng.uniform(min, max))


def random_scaling(min, max, prng=DEFAULT_PRNG):
    return scaling(_random_vector(min, max, prng))


def random_flip(flip_x_chance, flip_y_chance, prng=DEFAULT_PRNG):
    fx = prng.uniform(0, 1) < flip_x_chance
    fy = prng.uniform(0, 1) < flip_y_chance
    return scaling((1 - 2 * fx, 1 - 2 * fy))


def change_transform_origin(transform, center):
    """translate(center) . transform . translate(-center) (model/transform.py:177-187)."""
    center = np.array(center)
    return np.linalg.multi_dot([translation(center), transform, translation(-center)])


def random_transform(min_rotation=0, max_rotation=0, min_translation=(0, 0), max_translation=(0, 0), min_shear=0, max_shear=0,
                     min_scaling=(1, 1), max_scaling=(1, 1), flip_x_chance=0, flip_y_chance=0, prng=DEFAULT_PRNG):
    """rotation . translation . shear . scaling . flip, each drawn from `prng` in that order (model/transform.py:190-234)."""
    parts = [random_rotation(min_rotation, max_rotation, prng),
             random_translation(min_translation, max_translation, prng),
             random_shear(min_shear, max_shear, prng),
             random_scaling(min_scaling, max_scaling, prng),
             random_flip(flip_x_chance, flip_y_chance, prng)]
    return np.linalg.multi_dot(parts)


def random_transform_generator(prng=None, **kwargs):
    """Endless stream of random_transform(**kwargs) from one dedicated PRNG (model/transform.py:237-272)."""
    prng = np.random.RandomState() if prng is None else prng
    while True:
        yield random_transform(prng=prng, **kwargs)


def transform_aabb(transform, aabb):
    """Axis-aligned box around the four transformed corners (model/transform.py:17-42)."""
    x1, y1, x2, y2 = aabb
    pts = transform.dot([[x1, x2, x1, x2],
                         [y1, y2, y2, y1],
                         [1, 1, 1, 1]])
    lo, hi = pts.min(axis=1), pts.max(axis=1)
    return [lo[0], lo[1], hi[0], hi[1]]


class TransformParameters:
    """How apply_transform samples (model/transform.py:280-321): fill_mode 'constant' | 'nearest' | 'reflect' | 'wrap',
    interpolation 'nearest' | 'linear' (the device kernel implements these two; 'cubic', 'area', 'lanczos4' raise),
    cval for 'constant', relative_translation."""

    def __init__(self, fill_mode='nearest', interpolation='linear', cval=0, relative_translation=True):
        self.fill_mode = fill_mode
        self.cval = cval
        self.interpolation = interpolation
        self.relative_translation = relative_translation

    def cvBorderMode(self):
        return _BORDER_CV.get(self.fill_mode)

    def cvInterpolation(self):
        return _INTER_CV.get(self.interpolation)


def adjust_transform_for_image(transform, image, relative_translation):
    """Scale the translation by the image size (in place, as the reference does) and move the origin to the image centre
    (model/transform.py:324-340)."""
    height, width = image.shape[0], image.shape[1]
    if relative_translation:
        transform[0:2, 2] *= [width, height]
    return change_transform_origin(transform, (0.5 * width, 0.5 * height))


def invert_affine(matrix):
    """The destination->source map cv2.warpAffine derives from a forward 2x3 matrix (its own double-precision formula, recalled
    from OpenCV's imgwarp.cpp - not the general matrix inverse, so the roundings match)."""
    M = np.array(matrix, np.float64)[:2, :].copy()
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    M[0, 0] = A11
    M[0, 1] *= -D
    M[1, 0] *= -D
    M[1, 1] = A22
    b1 = -M[0, 0] * M[0, 2] - M[0, 1] * M[1, 2]
    b2 = -M[1, 0] * M[0, 2] - M[1, 1] * M[1, 2]
    M[0, 2], M[1, 2] = b1, b2
    return M


def warp_codes(params):
    """(interpolation, border_mode, cval4) of rtn_warp_affine_u8 for a TransformParameters."""
    inter, border = params.cvInterpolation(), params.cvBorderMode()
    if inter not in (0, 1):
        raise NotImplementedError("apply_transform: interpolation %r is not implemented on the device (nearest, linear are)" % (params.interpolation,))
    if border not in _BORDER_RTN:
        raise ValueError("apply_transform: unknown fill_mode %r" % (params.fill_mode,))
    # cv2 receives borderValue = params.cval as Scalar(cval): the first channel gets cval, the others 0
    cval = np.zeros(4, np.uint8)
    cval[0] = int(np.clip(np.rint(params.cval), 0, 255))
    return inter, _BORDER_RTN[border], cval


def apply_transform(matrix, image, params):
    """cv2.warpAffine(image, matrix[:2], dsize = image size, flags, borderMode, borderValue) on the device.  `image`: uint8 (H,W,C)
    NumPy array (returns NumPy) or CUDA tensor (returns a CUDA tensor, no host round trip)."""
    import ctypes
    import torch
    from . import _rt
    L = _rt.L
    inter, border, cval = warp_codes(params)
    on_device = isinstance(image, torch.Tensor)
    if (image.dtype != torch.uint8) if on_device else (np.asarray(image).dtype != np.uint8):
        raise ValueError("apply_transform: the device path takes uint8 pages (what cv2.imread hands the generator)")
    src = image.contiguous() if on_device else _rt.dev(image, torch.uint8)
    if src.dim() == 2:
        src = src[..., None]
    H, W, C = src.shape
    inv = np.ascontiguousarray(invert_affine(matrix))
    dst = torch.empty_like(src)
    h = _rt.handle()
    h.check(L.lib.rtn_warp_affine_u8(h.raw, src.data_ptr(), H, W, C, inv.ctypes.data_as(ctypes.c_void_p), inter, border,
                                     cval.ctypes.data_as(ctypes.c_void_p), dst.data_ptr()))
    if image.ndim == 2:
        dst = dst[..., 0]
    return dst if on_device else _rt.host(dst)
