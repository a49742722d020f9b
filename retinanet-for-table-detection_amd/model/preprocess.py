"""Page preprocessing of the reference on the device: DetectTablesUtils.preProcessSampleImages (DetectTablesUtils.py:229-261)
and utils.resize_image's cv2.resize (model/utils.py:152), plus Generator.compute_inputs' zero-padded batch
(csv_generator.py:320-336)."""
import numpy as np
import torch

from . import _rt

L = _rt.L


def preprocess_pages(pages, return_binary=False):
    """pages: uint8 (B,H,W,3) BGR or (B,H,W) gray, or one (H,W[,3]) page.  Returns uint8 (B,H,W,3): b=L2, g=L1, r=C distance
    maps of the Gaussian adaptive threshold, saturated to uint8 as cv2.imwrite stores them."""
    a = np.asarray(pages)
    single = a.ndim == 2 or (a.ndim == 3 and a.shape[-1] == 3)        # one gray (H,W) or one BGR (H,W,3) page
    if single:
        a = a[None]
    ch = 3 if a.ndim == 4 else 1
    if a.dtype != np.uint8:
        raise ValueError("pages must be uint8")
    B, H, W = a.shape[:3]
    h = _rt.handle()
    src = _rt.dev(a, torch.uint8)
    dst = torch.empty(B, H, W, 3, dtype=torch.uint8, device="cuda")
    binary = torch.empty(B, H, W, dtype=torch.uint8, device="cuda")
    wsb = L.lib.rtn_preprocess_dt3_workspace_bytes(B, H, W)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    h.check(L.lib.rtn_preprocess_dt3(h.raw, src.data_ptr(), ch, B, H, W, dst.data_ptr(), binary.data_ptr(), ws.data_ptr(), wsb))
    out, bo = _rt.host(dst), _rt.host(binary)
    if single:
        out, bo = out[0], bo[0]
    return (out, bo) if return_binary else out


def resize_cubic(img, scale):
    """cv2.resize(img, None, fx=scale, fy=scale, interpolation=cv2.INTER_CUBIC) of a float32 (H,W,C) image."""
    img = np.asarray(img, np.float32)
    if img.ndim == 2:
        return resize_cubic(img[..., None], scale)[..., 0]
    H, W, Cc = img.shape
    Ho, Wo = int(np.rint(H * scale)), int(np.rint(W * scale))
    h = _rt.handle()
    s = _rt.dev(img, torch.float32)
    out = torch.empty(Ho, Wo, Cc, dtype=torch.float32, device="cuda")
    h.check(L.lib.rtn_resize_cubic(h.raw, s.data_ptr(), L.RTN_F32, H, W, Cc, float(scale), out.data_ptr(), L.RTN_F32, Ho, Wo, Wo * Cc))
    return _rt.host(out)


def compute_inputs_device(processed_u8_pages, min_side=800, max_side=1333, dtype=torch.bfloat16):
    """csv_generator.Generator: preprocess_group + compute_inputs on the device.  processed_u8_pages: list of uint8 (H,W,3)
    distance-map pages (any sizes).  Normalises (x/127.5-1), resizes each page by its own scale (INTER_CUBIC) and writes it
    into the top-left of a zero canvas of the largest resized shape.  Returns (device tensor (B,Hmax,Wmax,3), scales)."""
    from .utils import compute_resize_scale
    h = _rt.handle()
    scales = [compute_resize_scale(p.shape, min_side, max_side) for p in processed_u8_pages]
    shapes = [(int(np.rint(p.shape[0] * s)), int(np.rint(p.shape[1] * s))) for p, s in zip(processed_u8_pages, scales)]
    Hm, Wm = max(s[0] for s in shapes), max(s[1] for s in shapes)
    canvas = torch.zeros(len(shapes), Hm, Wm, 3, dtype=dtype, device="cuda")
    code = L.RTN_BF16 if dtype == torch.bfloat16 else L.RTN_F32
    keep = []
    for i, (p, s, (ho, wo)) in enumerate(zip(processed_u8_pages, scales, shapes)):
        src = _rt.dev(p, torch.uint8)
        keep.append(src)
        h.check(L.lib.rtn_resize_cubic(h.raw, src.data_ptr(), 2, p.shape[0], p.shape[1], 3, float(s), canvas[i].data_ptr(), code, ho, wo, Wm * 3))
    torch.cuda.synchronize()
    return canvas, scales
