"""CPU test over the BUILT library's machine code: inside the MFMA blocks of the persistent convolution kernels (generations 4, 5, 7 and
the window weight-gradient kernel) the compiler may place no register moves, no `s_nop N >= 4`, no scratch access and no `vmcnt(0)`.
Round 3 found the fp8 tower kernel 25 % slower than necessary for exactly that reason (8-dword MFMA operands built element by element:
`v_pk_mov_b32` shuffles + `s_nop 6` in front of half of its MFMAs, profiles/r3_fp8_operand_fix.txt); tools/lint_mfma_blocks.py keeps
the class out after any rebuild.  Also checks that the lint still recognises that pattern (a synthetic listing)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("lint_mfma_blocks", os.path.join(ROOT, "tools", "lint_mfma_blocks.py"))
lint = importlib.util.module_from_spec(spec)
spec.loader.exec_module(lint)

STRICT = r"conv_halo8_kernel|conv_gemm8_kernel|conv_wgrad_win_kernel"


def test_lint_flags_the_round3_fp8_sequence():
    mf = "v[240:243], v[196:203], v[26:33], v[22:25], v173, v173"
    ins = [(0x00, "v_mfma_scale_f32_16x16x128_f8f6f4", mf),
           (0x08, "s_nop", "6"),
           (0x0c, "v_pk_mov_b32", "v[22:23], v[34:35], v[36:37] op_sel:[1,0]"),
           (0x14, "v_mfma_scale_f32_16x16x128_f8f6f4", mf),
           (0x1c, "v_mov_b32_e32", "v35, v22"),
           (0x20, "s_waitcnt", "lgkmcnt(2)"),
           (0x24, "v_mfma_scale_f32_16x16x128_f8f6f4", mf),
           (0x2c, "s_endpgm", "")]
    r = lint.lint_kernel(ins)
    assert r["mfma"] == 3 and r["moves"] == 2 and r["long_nops"] == 1 and r["scratch"] == 0 and r["vmcnt0"] == 0
    # the clean form: waits and fragment reads between MFMAs are what the schedule puts there
    ins = [(0x00, "v_mfma_f32_16x16x32_bf16", "v[0:3], v[4:7], v[8:11], v[0:3]"),
           (0x08, "s_waitcnt", "lgkmcnt(1)"),
           (0x0c, "ds_read_b128", "v[8:11], v20 offset:2048"),
           (0x14, "s_nop", "1"),
           (0x18, "v_mfma_f32_16x16x32_bf16", "v[12:15], v[4:7], v[8:11], v[12:15]")]
    r = lint.lint_kernel(ins)
    assert r["mfma"] == 2 and not (r["moves"] or r["long_nops"] or r["scratch"] or r["vmcnt0"])
    # two MFMAs far apart are two blocks: what lies between them is not judged
    ins = [(0, "v_mfma_f32_16x16x32_bf16", "")] + [(4 + 4 * i, "v_mov_b32_e32", "v1, v2") for i in range(lint.GAP + 2)] + [(200, "v_mfma_f32_16x16x32_bf16", "")]
    assert lint.lint_kernel(ins)["moves"] == 0


def test_the_mfma_blocks_of_librtn_are_clean(pkg):
    if not os.path.exists(os.path.join(lint._scan.LLVM, "llvm-objdump")):
        pytest.skip("no llvm-objdump in this image")
    path = pkg._lib.LIB_PATH
    rows = lint.lint_file(path, STRICT)
    assert len(rows) >= 40, "expected the persistent kernels' instances in %s, found %d" % (path, len(rows))
    bad = {k: v for k, v in rows.items() if v["moves"] or v["long_nops"] or v["scratch"] or v["vmcnt0"]}
    assert not bad, "MFMA blocks with compiler artefacts: %s" % {k[:80]: v["examples"][:2] for k, v in bad.items()}
