"""Lint the MFMA blocks of a gfx950 code object: inside a run of MFMAs of the persistent convolution kernels nothing but MFMAs (and the
waits / fragment reads their schedule places there) may be issued.

Background (profiles/r3_fp8_operand_fix.txt): the fp8 instance of the head-tower kernel built its 8-dword MFMA operands element by
element; LLVM turned that into `v_pk_mov_b32` + `v_mov_b32` shuffles and an `s_nop 6` (the VALU-write -> MFMA-read hazard) in front of
half of the block's MFMAs, and the layer ran 25 % slower than it had to - found only by reading the ISA.  This scan keeps that class of
compiler artefact out: for every kernel whose name matches --kernels it walks the instruction stream, takes two MFMAs at most GAP
instructions apart as one block, and counts between them register moves (v_mov / v_pk_mov / v_accvgpr), `s_nop N` with N >= 4,
scratch accesses and `s_waitcnt vmcnt(0)`.

usage: python tools/lint_mfma_blocks.py [--kernels REGEX] [--json] <librtn.so | file.o | code-object> ...
exit status 1 when a matching kernel has such an instruction inside an MFMA block."""
import argparse
import importlib.util
import json
import os
import re
import sys
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("scan_store_hazard", os.path.join(_HERE, "scan_store_hazard.py"))
_scan = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_scan)

GAP = 10
DEFAULT_KERNELS = r"conv_halo8_kernel|conv_gemm8_kernel|conv_halon_kernel|conv_wgrad_win_kernel"
_nop = re.compile(r"^(\d+)")


def lint_kernel(ins):
    """ins: [(address, mnemonic, operands)].  Returns counts of offending instructions inside MFMA blocks."""
    idx = [i for i, (_, m, _) in enumerate(ins) if m.startswith("v_mfma")]
    bad = {"moves": 0, "long_nops": 0, "scratch": 0, "vmcnt0": 0}
    where = []
    for a, b in zip(idx, idx[1:]):
        if b - a > GAP:
            continue
        for addr, m, ops in ins[a + 1:b]:
            kind = None
            if m.startswith(("v_mov_b32", "v_mov_b64", "v_pk_mov", "v_accvgpr")):
                kind = "moves"
            elif m == "s_nop":
                mm = _nop.match(ops.strip())
                if mm and int(mm.group(1)) >= 4:
                    kind = "long_nops"
            elif m.startswith("scratch_"):
                kind = "scratch"
            elif m == "s_waitcnt" and "vmcnt(0)" in ops:
                kind = "vmcnt0"
            if kind:
                bad[kind] += 1
                if len(where) < 8:
                    where.append("%x: %s %s" % (addr, m, ops))
    return {"mfma": len(idx), **bad, "examples": where}


def lint_file(path, pattern):
    rx = re.compile(pattern)
    rows = {}
    with tempfile.TemporaryDirectory() as tmp:
        for co in _scan.code_objects(path, tmp):
            for name, ins in _scan.disassemble(co).items():
                if rx.search(name):
                    r = lint_kernel(ins)
                    if r["mfma"]:
                        rows[name] = r
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("paths", nargs="+")
    ap.add_argument("--kernels", default=DEFAULT_KERNELS)
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    failed = False
    out = {}
    for p in a.paths:
        rows = lint_file(p, a.kernels)
        out[p] = rows
        for name, r in sorted(rows.items()):
            n = r["moves"] + r["long_nops"] + r["scratch"] + r["vmcnt0"]
            failed = failed or n > 0
            if not a.json:
                print("%-110s mfma %4d  moves %d  s_nop>=4 %d  scratch %d  vmcnt(0) %d%s" % (
                    name[:110], r["mfma"], r["moves"], r["long_nops"], r["scratch"], r["vmcnt0"], ("   e.g. " + r["examples"][0]) if n else ""))
    if a.json:
        print(json.dumps(out, indent=1))
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
