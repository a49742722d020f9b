"""CPU tests of the drop-in boundary: librtn.so loads, exports every symbol include/rtn.h declares,
the ctypes structs match the header's layout, and host-only entry points give reference answers.
No kernel is launched here."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rtn.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtn_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    syms = declared_symbols()
    assert len(syms) >= 18
    lib = C.CDLL(pkg.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), "librtn.so does not export %s" % s
    assert set(syms) == set(pkg._lib.SIGNATURES), "ctypes binding and header disagree"


def test_struct_layout_matches_header(pkg, tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "rtn.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(rtn_conv_group_t),sizeof(rtn_conv_desc_t),sizeof(rtn_anchor_cfg_t),'
                   'offsetof(rtn_conv_desc_t,w),offsetof(rtn_conv_desc_t,flags),offsetof(rtn_anchor_cfg_t,base),'
                   'sizeof(rtn_conv_src2_t),offsetof(rtn_conv_src2_t,step),'
                   'offsetof(rtn_conv_desc_t,workspace),offsetof(rtn_conv_desc_t,workspace_bytes));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    L = pkg._lib
    want = [C.sizeof(L.ConvGroup), C.sizeof(L.ConvDesc), C.sizeof(L.AnchorCfg), L.ConvDesc.w.offset, L.ConvDesc.flags.offset,
            L.AnchorCfg.base.offset, C.sizeof(L.ConvSrc2), L.ConvSrc2.step.offset,
            L.ConvDesc.workspace.offset, L.ConvDesc.workspace_bytes.offset]
    assert got == want


def test_generate_anchors_host_entry(pkg, golden):
    from oracle import ref_numpy as R
    for size in (32, 64, 128, 256, 512, 48):
        got = pkg._lib.generate_anchors_f64(size, R.DEFAULT_RATIOS, R.DEFAULT_SCALES)
        assert np.array_equal(got, golden["base_%d" % size])
    # float64 scales take the float64 product path, like NumPy would
    s64 = np.array([1.0, 2 ** (1 / 3), 2 ** (2 / 3)])
    assert np.array_equal(pkg._lib.generate_anchors_f64(48, R.DEFAULT_RATIOS, s64), R.base_anchors(48, R.DEFAULT_RATIOS, s64))


def test_error_paths_without_gpu(pkg):
    L = pkg._lib
    assert L.lib.rtn_version().startswith(b"librtn")
    assert L.lib.rtn_destroy(None) == -1
    assert L.lib.rtn_conv2d_fwd(None, None) == -1
    assert L.lib.rtn_detect_workspace_bytes(0, 10, 1) == 0
    assert L.lib.rtn_detect_workspace_bytes(2, 1000, 1) > 2 * 1000 * 8
    bad = (C.c_double * 1)(1.0)
    assert L.lib.rtn_generate_anchors(32.0, bad, 0, bad, 1, bad) == -1


def test_wgrad_workspace_sizes_without_gpu(pkg, monkeypatch):
    """rtn_conv2d_wgrad_workspace_bytes is a host function: the row-info table (16 B per 64-padded pixel) plus, unless RTN_WGRAD_SLAB=0,
    the per-split slabs of the ordered (atomic-free) reduction, or the nine-tap window kernel's slabs where that kernel is taken - always BEHIND the table."""
    L = pkg._lib

    def desc(H, W, cin, cout, k, B=8):
        d = L.ConvDesc()
        d.ngroups, d.batch, d.dtype = 1, B, L.RTN_BF16
        d.w_rows, d.N, d.KH, d.KW = cout, cout, k, k
        d.Crun = d.pix_stride = cin
        d.sy = d.sx = 1
        d.pad_t = d.pad_l = k // 2
        d.out_ld = cout
        g = d.g[0]
        g.Hin, g.Win, g.Hout, g.Wout = H, W, H, W
        g.in_row_stride, g.in_img_stride, g.out_img_stride = W * cin, H * W * cin, H * W * cout
        g.in_elems, g.out_elems = B * H * W * cin, B * H * W * cout
        return d

    d = desc(50, 84, 256, 64, 1)
    table = ((8 * 50 * 84 + 63) // 64) * 64 * 16
    monkeypatch.setenv("RTN_WGRAD_SLAB", "0")
    assert table <= L.lib.rtn_conv2d_wgrad_workspace_bytes(C.byref(d)) < table + 256
    monkeypatch.setenv("RTN_WGRAD_SLAB", "1")
    with_slabs = L.lib.rtn_conv2d_wgrad_workspace_bytes(C.byref(d))
    assert with_slabs >= table + 8 * 64 * (256 + 1) * 4             # at least 8 pixel splits of [N][K] + [N] floats
    assert (with_slabs - table) % (64 * 257 * 4) < 256               # a whole number of split slabs behind the (aligned) table
    d3 = desc(50, 84, 256, 256, 3)                                   # res4 branch2b
    assert L.lib.rtn_conv2d_wgrad_workspace_bytes(None) == 0
    # the nine-tap window kernel (csrc/rtn_wgrad_win.hip): S slabs of [N][9 C] accumulator fragments + S x (C / 64) bias parts of [N],
    # S = 8 x 32 / output tiles; taken by work (RTN_WGRAD_WIN unset), wherever the shape allows (1), never (0)
    monkeypatch.setenv("RTN_WGRAD_WIN", "0")
    general = L.lib.rtn_conv2d_wgrad_workspace_bytes(C.byref(d3))
    table3 = (((8 * 50 * 84 + 63) // 64) * 64 * 16 + 255) // 256 * 256
    monkeypatch.setenv("RTN_WGRAD_WIN", "1")
    assert L.lib.rtn_conv2d_wgrad_workspace_bytes(C.byref(d3)) == table3 + 32 * 256 * (9 * 256 + 4) * 4 > general      # 8 output tiles, 32 splits
    monkeypatch.delenv("RTN_WGRAD_WIN")
    assert L.lib.rtn_conv2d_wgrad_workspace_bytes(C.byref(d3)) == table3 + 32 * 256 * (9 * 256 + 4) * 4       # res4 branch2b is taken by default
    d2 = desc(200, 334, 64, 64, 3)                                   # res2 branch2b: the 64-filter form, one output tile, 256 splits
    assert L.lib.rtn_conv2d_wgrad_workspace_bytes(C.byref(d2)) >= 256 * 64 * (9 * 64 + 1) * 4
    small = desc(25, 42, 256, 256, 3)                                # P5: too little work for a chip-wide grid, stays on the general kernel
    by_default = L.lib.rtn_conv2d_wgrad_workspace_bytes(C.byref(small))
    monkeypatch.setenv("RTN_WGRAD_WIN", "0")
    assert L.lib.rtn_conv2d_wgrad_workspace_bytes(C.byref(small)) == by_default
    wide = desc(8, 400, 128, 128, 3)                                 # image rows of 400 pixels: the ring does not fit the LDS, never taken
    monkeypatch.setenv("RTN_WGRAD_WIN", "1")
    forced = L.lib.rtn_conv2d_wgrad_workspace_bytes(C.byref(wide))
    monkeypatch.setenv("RTN_WGRAD_WIN", "0")
    assert L.lib.rtn_conv2d_wgrad_workspace_bytes(C.byref(wide)) == forced


def test_product_does_not_import_oracle():
    pkgdir = os.path.join(ROOT, "retinanet-for-table-detection_amd")
    for dirpath, _, files in os.walk(pkgdir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f


def test_one_hip_runtime_in_the_process():
    """librtn.so must bind to the HIP runtime PyTorch loaded (same SONAME), not bring /opt/rocm's copy in beside it: with two
    runtimes in one process, streams and pointers cross between them and device discovery fails intermittently.  Checked in a
    fresh interpreter that imports the package FIRST (the order __graft_entry__.build() uses)."""
    code = ("import importlib, sys; sys.path.insert(0, %r); importlib.import_module('retinanet-for-table-detection_amd'); import torch; "
            "print(sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l)))" % ROOT)
    out = subprocess.check_output([sys.executable, "-c", code], text=True).strip().splitlines()[-1]
    libs = eval(out)
    assert len(libs) == 1, libs
