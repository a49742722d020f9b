"""CPU test over the BUILT library's machine code: no store of >= 96 bits may have its data VGPRs rewritten within fewer than two
wait states (the documented gfx940+ distance).  LLVM leaves buffer stores with an SGPR soffset unpadded, which is what corrupted
the fused bottleneck block's output in round 2 (profiles/r3_store_hazard_isa.txt); the kernels guard such stores explicitly and
this scan keeps every other kernel file honest after any rebuild.  Also checks that the scanner still recognises the failing
pattern (a synthetic listing with the round-2 instruction sequence)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("scan_store_hazard", os.path.join(ROOT, "tools", "scan_store_hazard.py"))
scan = importlib.util.module_from_spec(spec)
spec.loader.exec_module(scan)


def test_scanner_flags_the_round2_sequence():
    ins = [(0x100, "v_cvt_pk_bf16_f32", "v5, v5, v10"),
           (0x108, "buffer_store_dwordx4", "v[2:5], v132, s[40:43], s62 offen"),
           (0x110, "s_waitcnt", "vmcnt(11)"),
           (0x114, "v_lshlrev_b32_e32", "v2, 16, v34"),
           (0x118, "s_endpgm", "")]
    rows = scan.scan_kernel("k", ins)
    assert len(rows) == 1 and rows[0]["wait_states"] == 1 and rows[0]["sgpr_soffset"] and not rows[0]["async_writer"]
    # the guarded form: s_nop 3 (4 wait states) directly behind the store
    ins[2] = (0x110, "s_nop", "3")
    assert scan.scan_kernel("k", ins)[0]["wait_states"] == 4
    # a rewrite reached through a taken branch is found as well
    ins = [(0x100, "global_store_dwordx4", "v[8:9], v[4:7], off"),
           (0x108, "s_cbranch_scc1", "2"),                      # -> 0x108 + 4 + 8 = 0x114
           (0x10c, "s_nop", "7"),
           (0x110, "s_endpgm", ""),
           (0x114, "v_mov_b32_e32", "v6, 0"),
           (0x118, "s_endpgm", "")]
    rows = scan.scan_kernel("k", ins)
    assert rows and rows[0]["wait_states"] == 1 and rows[0]["writer"].startswith("v_mov_b32")


def test_no_store_in_librtn_is_rewritten_below_the_documented_distance(pkg):
    if not os.path.exists(os.path.join(scan.LLVM, "llvm-objdump")):
        pytest.skip("llvm-objdump not found under %s" % scan.LLVM)
    rows, nstores = scan.scan([pkg.LIB_PATH])
    assert nstores > 500, "the scan saw only %d wide stores: disassembly failed?" % nstores
    bad = [r for r in rows if not r["async_writer"] and r["wait_states"] < 2]
    assert not bad, "stores whose data registers are rewritten too early:\n" + "\n".join(
        "%s +%s: %s <- %s after %d wait state(s)" % (r["kernel"], r["addr"], r["store"], r["writer"], r["wait_states"]) for r in bad)
