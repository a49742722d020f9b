"""ctypes binding of librtn.so (the C-ABI declared in include/rtn.h).

There is no CPU fallback: if the shared library is missing or a symbol is absent the import
raises, and every call checks its return code and raises RtnError with rtn_last_error().
"""
import ctypes as C
import os

# PyTorch must load ITS HIP runtime before librtn.so is opened: torch bundles libamdhip64.so (SONAME libamdhip64.so.7) and
# librtn.so needs libamdhip64.so.7, so opened second it binds to the copy already in the process.  Opened first, the loader
# takes /opt/rocm/lib/libamdhip64.so.7 and torch then brings a second runtime in: two HIP runtimes in one process hand each
# other streams and pointers, and device discovery in the second one fails intermittently ("no ROCm-capable device").
import torch  # noqa: F401  (load order, see above)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTN_LIB_PATH") or os.path.join(_HERE, "librtn.so")      # RTN_LIB_PATH: A/B two builds on one box

RTN_BF16, RTN_F32, RTN_U8, RTN_FP8 = 0, 1, 2, 3
RTN_MAX_GROUPS, RTN_MAX_GT, RTN_MAX_DET = 5, 64, 300

CONV_RELU, CONV_SIGMOID, CONV_RES_SAME, CONV_RES_UPSAMPLE, CONV_OUT_F32, CONV_RELU_MASK, CONV_MASK_PRE = 0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40

ERRNAMES = {0: "RTN_OK", -1: "RTN_EINVAL", -2: "RTN_EHIP", -3: "RTN_ENOMEM", -4: "RTN_EBOUNDS"}


class RtnError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("%s (%d): %s" % (ERRNAMES.get(code, "RTN_E?"), code, text))
        self.code = code


class ConvGroup(C.Structure):
    _fields_ = [
        ("in_", C.c_void_p), ("out", C.c_void_p), ("res", C.c_void_p),
        ("in_elems", C.c_int64), ("out_elems", C.c_int64), ("res_elems", C.c_int64),
        ("in_img_stride", C.c_int64), ("out_img_stride", C.c_int64), ("out_off", C.c_int64),
        ("res_img_stride", C.c_int64),
        ("in_row_stride", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32),
        ("Hout", C.c_int32), ("Wout", C.c_int32), ("Hres", C.c_int32), ("Wres", C.c_int32),
        ("res_ld", C.c_int32),
        ("mask", C.c_void_p), ("mask_elems", C.c_int64), ("mask_img_stride", C.c_int64),
        ("mask_ld", C.c_int32), ("out_step", C.c_int32), ("out_pix_w", C.c_int32), ("reserved_", C.c_int32),
    ]


class ConvDesc(C.Structure):
    _fields_ = [
        ("g", ConvGroup * RTN_MAX_GROUPS),
        ("ngroups", C.c_int32), ("batch", C.c_int32), ("dtype", C.c_int32),
        ("w", C.c_void_p), ("bias", C.c_void_p),
        ("w_rows", C.c_int32), ("N", C.c_int32), ("KH", C.c_int32), ("KW", C.c_int32),
        ("Crun", C.c_int32), ("pix_stride", C.c_int32), ("sy", C.c_int32), ("sx", C.c_int32),
        ("pad_t", C.c_int32), ("pad_l", C.c_int32), ("out_ld", C.c_int32), ("flags", C.c_int32),
        ("reserved_", C.c_int32), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
    ]


class ConvSrc2(C.Structure):
    _fields_ = [
        ("in_", C.c_void_p), ("in_elems", C.c_int64), ("in_img_stride", C.c_int64),
        ("in_row_stride", C.c_int32), ("pix_stride", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32),
        ("C", C.c_int32), ("step", C.c_int32),
    ]


class AnchorCfg(C.Structure):
    _fields_ = [
        ("nlevels", C.c_int32), ("A", C.c_int32),
        ("H", C.c_int32 * RTN_MAX_GROUPS), ("W", C.c_int32 * RTN_MAX_GROUPS),
        ("stride", C.c_int32 * RTN_MAX_GROUPS),
        ("anchor_off", C.c_int32 * (RTN_MAX_GROUPS + 1)),
        ("base", ((C.c_double * 4) * 16) * RTN_MAX_GROUPS),
    ]


class BottleneckDesc(C.Structure):
    _fields_ = [
        ("a_in", C.c_void_p), ("a_in_elems", C.c_int64), ("x_in", C.c_void_p), ("x_in_elems", C.c_int64),
        ("x_out", C.c_void_p), ("x_out_elems", C.c_int64), ("a_out", C.c_void_p), ("a_out_elems", C.c_int64),
        ("w2b", C.c_void_p), ("b2b", C.c_void_p), ("w2c", C.c_void_p), ("b2c", C.c_void_p), ("w2a", C.c_void_p), ("b2a", C.c_void_p),
        ("batch", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("mid", C.c_int32), ("dtype", C.c_int32), ("w2c_ld", C.c_int32),
        ("p_in", C.c_void_p), ("p_in_elems", C.c_int64), ("wproj", C.c_void_p),
        ("h1_out", C.c_void_p), ("h1_out_elems", C.c_int64),
    ]


class ChainDesc(C.Structure):
    _fields_ = [
        ("h_in", C.c_void_p), ("h_in_elems", C.c_int64), ("x_in", C.c_void_p), ("x_in_elems", C.c_int64),
        ("x_out", C.c_void_p), ("x_out_elems", C.c_int64), ("a_out", C.c_void_p), ("a_out_elems", C.c_int64),
        ("w2c", C.c_void_p), ("b2c", C.c_void_p), ("w2a", C.c_void_p), ("b2a", C.c_void_p),
        ("pixels", C.c_int64), ("mid", C.c_int32), ("out", C.c_int32), ("next", C.c_int32), ("dtype", C.c_int32),
    ]


class ConvFp8(C.Structure):
    _fields_ = [("acc_scale", C.c_float), ("out_scale", C.c_float), ("out_dtype", C.c_int32)]


# every symbol include/rtn.h declares: (restype, argtypes)
_P, _I, _I64, _F, _D, _SZ = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_size_t
SIGNATURES = {
    "rtn_create": (_I, [C.POINTER(_P), _I]),
    "rtn_destroy": (_I, [_P]),
    "rtn_create_error": (C.c_char_p, []),
    "rtn_set_stream": (_I, [_P, _P]),
    "rtn_last_error": (C.c_char_p, [_P]),
    "rtn_version": (C.c_char_p, []),
    "rtn_conv2d_workspace_bytes": (_SZ, [_P, C.POINTER(ConvDesc)]),
    "rtn_debug_last_conv_impl": (_I, [_P]),
    "rtn_debug_last_wgrad_impl": (_I, [_P]),
    "rtn_conv2d_fwd": (_I, [_P, C.POINTER(ConvDesc)]),
    "rtn_conv1x1_dual_fwd": (_I, [_P, C.POINTER(ConvDesc), C.POINTER(ConvSrc2)]),
    "rtn_conv1x1_dual_workspace_bytes": (_SZ, [_P, C.POINTER(ConvDesc), C.POINTER(ConvSrc2)]),
    "rtn_conv_workspace_init": (_I, [_P, _P, _SZ]),
    "rtn_debug_last_conv_streamk": (_I, [_P]),
    "rtn_debug_last_conv_tile": (_I, [_P]),
    "rtn_debug_conv_sync_timeouts": (_I, [_P, _P, C.POINTER(C.c_uint32)]),
    "rtn_bottleneck64_fwd": (_I, [_P, C.POINTER(BottleneckDesc)]),
    "rtn_chain1x1_fwd": (_I, [_P, C.POINTER(ChainDesc)]),
    "rtn_chain1x1_supported": (_I, [_I, _I, _I]),
    "rtn_conv2d_dgrad": (_I, [_P, C.POINTER(ConvDesc)]),
    "rtn_pack_dgrad_weights": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I]),
    "rtn_pack_dgrad_weights_multi": (_I, [_P, _P, _I, _I64, _I]),
    "rtn_conv2d_wgrad_workspace_bytes": (_SZ, [C.POINTER(ConvDesc)]),
    "rtn_conv2d_wgrad": (_I, [_P, C.POINTER(ConvDesc), _P, _P, _SZ]),
    "rtn_conv2d_wgrad_bias": (_I, [_P, C.POINTER(ConvDesc), _P, _P, _I, _P, _SZ]),
    "rtn_conv2d_wgrad_rowinfo": (_I, [_P, C.POINTER(ConvDesc), _P, _SZ]),
    "rtn_conv2d_wgrad_prepared": (_I, [_P, C.POINTER(ConvDesc), _P, _P, _I, _P, _SZ]),
    "rtn_bias_grad": (_I, [_P, _P, _I, _I64, _I, _I64, _P]),
    "rtn_pad_cast_rows": (_I, [_P, _P, _P, _I, _I64, _I, _I]),
    "rtn_zero_insert2": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I]),
    "rtn_upsample_add_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I]),
    "rtn_maxpool3x3s2_tfsame_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I]),
    "rtn_maxpool3x3s2_tfsame_fwd_idx": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I]),
    "rtn_maxpool3x3s2_tfsame_bwd_idx": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I]),
    "rtn_sumsq_workspace_bytes": (_SZ, []),
    "rtn_sumsq": (_I, [_P, _P, _P, _I64, _P, _P, _SZ]),
    "rtn_adam_clipnorm_step": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I64, _I64, _F, _F, _F, _F, _P, _F, _F]),
    "rtn_sumsq_segments": (_I, [_P, _P, _P, _P, _I, _P]),
    "rtn_adam_clipnorm_step_segments": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I64, _I64, _F, _F, _F, _F, _P, _I, _P, _I64, _F, _F]),
    "rtn_stem_pack": (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _I, _I]),
    "rtn_stem_conv_pool": (_I, [_P, _P, _I, _I, _P, _I, _P, _P, _I, _I, _I]),
    "rtn_stem_conv_pool_branch2a": (_I, [_P, _P, _I, _I, _P, _I, _P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "rtn_maxpool3x3s2_tfsame_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I]),
    "rtn_relu": (_I, [_P, _P, _P, _I, _I64]),
    "rtn_generate_anchors": (_I, [_D, C.POINTER(_D), _I, C.POINTER(_D), _I, C.POINTER(_D)]),
    "rtn_anchors_f64": (_I, [_P, C.POINTER(AnchorCfg), _P]),
    "rtn_anchors_f32": (_I, [_P, C.POINTER(AnchorCfg), _P]),
    "rtn_anchor_targets": (_I, [_P, C.POINTER(AnchorCfg), _I, _I, _P, _P, _P, _P, _D, _D, _P, _P]),
    "rtn_anchor_targets_explicit": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _D, _D, _P, _P]),
    "rtn_compute_overlap": (_I, [_P, _P, _P, _I, _I, _P]),
    "rtn_gt_annotations": (_I, [_P, _P, _I, _I, _D, _D, _P, _P, _P]),
    "rtn_bbox_transform": (_I, [_P, _P, _P, _I, C.POINTER(_D), C.POINTER(_D), _P]),
    "rtn_filter_detections": (_I, [_P, _I, _I64, _I, _P, _P, _F, _F, _I, _P, _P, _P, _P, _SZ]),
    "rtn_regress_boxes": (_I, [_P, _P, _P, _I64, C.POINTER(_F), C.POINTER(_F), _P]),
    "rtn_clip_boxes": (_I, [_P, _P, _I64, _F, _F, _P]),
    "rtn_upsample_nearest": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I]),
    "rtn_preprocess_image": (_I, [_P, _P, _I, _P, _I64, _I, _F, _F]),
    "rtn_preprocess_dt3_workspace_bytes": (_SZ, [_I, _I, _I]),
    "rtn_preprocess_dt3": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _SZ]),
    "rtn_distance_transform3": (_I, [_P, _P, _I, _I, _I, _P, _P, _SZ]),
    "rtn_resize_cubic": (_I, [_P, _P, _I, _I, _I, _I, _D, _P, _I, _I, _I, _I64]),
    "rtn_conv2d_fp8_fwd": (_I, [_P, C.POINTER(ConvDesc), C.POINTER(ConvFp8)]),
    "rtn_quantize_fp8": (_I, [_P, _P, _I, _P, _I64, _F]),
    "rtn_conv2d_fwd_fp8out": (_I, [_P, C.POINTER(ConvDesc), _F]),
    "rtn_warp_affine_u8": (_I, [_P, _P, _I, _I, _I, _P, _I, _I, _P, _P]),
    "rtn_retina_loss_fwd": (_I, [_P, _I64, _I, _P, _P, _P, _P, _F, _F, _F, _P, _P, _SZ]),
    "rtn_retina_loss_workspace_bytes": (_SZ, [_I64]),
    "rtn_retina_loss_bwd": (_I, [_P, _I64, _I, _P, _P, _P, _P, _F, _F, _F, _F, _F, _I, _P, _P]),
    "rtn_retina_loss_bwd_dev": (_I, [_P, _I64, _I, _P, _P, _P, _P, _F, _F, _F, _P, _I, _P, _P]),
    "rtn_detect_workspace_bytes": (_SZ, [_I, _I64, _I]),
    "rtn_decode_filter_nms": (_I, [_P, C.POINTER(AnchorCfg), _I, _I, _P, _P, _I, _I, _F, _F, _I, _P, _P, _P, _P, _SZ]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "librtn.so not found at %s — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C retinanet-for-table-detection_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class Handle:
    """Owns one rtn_handle_t bound to a device; check() turns error codes into RtnError."""

    def __init__(self, device=0):
        self._h = _P()
        rc = lib.rtn_create(C.byref(self._h), int(device))
        if rc != 0:
            raise RtnError(rc, "rtn_create(device=%d) failed: %s" % (device, lib.rtn_create_error().decode() or "no usable GPU?"))
        self.device = device

    @property
    def raw(self):
        return self._h

    def check(self, rc):
        if rc != 0:
            raise RtnError(rc, lib.rtn_last_error(self._h).decode("utf-8", "replace"))

    def set_stream(self, stream_ptr):
        self.check(lib.rtn_set_stream(self._h, _P(stream_ptr)))

    def close(self):
        if self._h:
            lib.rtn_destroy(self._h)
            self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def attach_conv_workspace(handle, d, s2=None):
    """Give a conv descriptor the scratch rtn_conv2d_workspace_bytes() (with `s2`: rtn_conv1x1_dual_workspace_bytes()) asks for on
    this device: a uint8 device tensor kept alive by the descriptor (the library never allocates; the K-split paths are skipped
    without it).  One buffer per descriptor, so launches on different streams never share scratch.  Its sync block (the first
    RTN_CONV_SYNC_BYTES) is zeroed here, once, on the handle's stream (rtn_conv_workspace_init).  Returns the tensor or None."""
    if s2 is None:
        n = int(lib.rtn_conv2d_workspace_bytes(handle.raw, C.byref(d)))
    else:
        n = int(lib.rtn_conv1x1_dual_workspace_bytes(handle.raw, C.byref(d), C.byref(s2)))
    if n <= 0:
        d.workspace, d.workspace_bytes, d._ws = None, 0, None
        return None
    t = torch.empty(n, dtype=torch.uint8, device=torch.device("cuda", handle.device))
    handle.check(lib.rtn_conv_workspace_init(handle.raw, t.data_ptr(), n))
    d.workspace, d.workspace_bytes, d._ws = t.data_ptr(), n, t
    return t


def generate_anchors_f64(base_size, ratios, scales):
    """Host-side generate_anchors (model/anchors.py:243-278) through the C-ABI."""
    import numpy as np
    r = np.ascontiguousarray(ratios, dtype=np.float64)
    s = np.ascontiguousarray(scales, dtype=np.float64)
    out = np.zeros((len(r) * len(s), 4), dtype=np.float64)
    rc = lib.rtn_generate_anchors(float(base_size), r.ctypes.data_as(C.POINTER(_D)), len(r),
                                  s.ctypes.data_as(C.POINTER(_D)), len(s), out.ctypes.data_as(C.POINTER(_D)))
    if rc != 0:
        raise RtnError(rc, "rtn_generate_anchors")
    return out
