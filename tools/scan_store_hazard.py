"""Scan gfx950 machine code for the VMEM store-data hazard: a buffer/global/flat store of more than 64 bits whose data VGPRs are
rewritten before the store has read them.

Background (profiles/r3_store_hazard_isa.txt): gfx940+ documents 2 wait states between such a store and a VALU write of its data
registers; LLVM's GCNHazardRecognizer::createsVALUHazard pads them with `s_nop 1` — EXCEPT for MUBUF/MTBUF stores whose soffset is
an SGPR, which it treats as immune (a rule inherited from the SI/CI documentation).  hipcc therefore emits
`buffer_store_dwordx4 v[2:5], v132, s[40:43], s62 offen ; s_waitcnt vmcnt(11) ; v_lshlrev_b32 v2, 16, v34` — ONE wait state — in
the unguarded fused-bottleneck kernel, the variant that produced run-to-run output corruption on MI355X (round 2, commit a70f70d).

The scan walks every kernel of a code object: for each store of >= 96 bits it follows the instruction stream (fall-through and
branch targets, up to WINDOW wait states) until an instruction writes one of the store's data VGPRs, and reports the number of
wait states in between (every instruction counts 1, `s_nop N` counts N + 1 — LLVM's model).  Writers that return asynchronously
(VMEM loads, DS reads, LDS-DMA) are listed separately: their write-back is hundreds of cycles away.

usage: python tools/scan_store_hazard.py [--min-wait N] [--json] <librtn.so | file.o | code-object> ...
exit status 1 when a store has fewer than --min-wait (default 2) wait states."""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = os.environ.get("RTN_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
WINDOW = 12

_sym = re.compile(r"^([0-9a-f]+) <([^>]+)>:")
_ins = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
_vreg = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
STORE = re.compile(r"^(buffer|global|flat|scratch)_store_(dwordx3|dwordx4|b96|b128)$")
ASYNC = re.compile(r"^(buffer_load|global_load|flat_load|scratch_load|ds_read|ds_load|ds_bpermute|ds_permute|ds_swizzle|ds_.*_rtn|buffer_atomic|global_atomic|flat_atomic|tbuffer_load|image_)")


def code_objects(path, tmp):
    """gfx950 ELF code objects inside `path` (a fat binary: shared library / object with offload bundles; or a bare code object)."""
    with open(path, "rb") as f:
        head = f.read(20)
    if head[:4] == b"\x7fELF" and head[18:20] == b"\xe0\x00":          # e_machine EM_AMDGPU (224)
        return [path]
    local = os.path.join(tmp, os.path.basename(path))
    shutil.copy(path, local)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    return sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if f.startswith(os.path.basename(path) + ".") and "amdgcn" in f)


def disassemble(co):
    out = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", co], check=True, capture_output=True, text=True).stdout
    kernels, cur = {}, None
    for line in out.splitlines():
        m = _sym.match(line)
        if m:
            cur = kernels.setdefault(m.group(2), [])
            continue
        m = _ins.match(line)
        if m and cur is not None:
            cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return kernels


def vregs(operand):
    out = set()
    for m in _vreg.finditer(operand):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def split_ops(ops):
    return [o.strip() for o in ops.split(",")] if ops else []


def written_vgprs(mn, ops):
    """VGPRs an instruction writes (destination = first operand of VALU / MFMA / loads; none for stores, scalar and v_cmp*)."""
    if mn.startswith("s_") or STORE.match(mn) or "_store_" in mn or mn.startswith("ds_write") or mn.startswith("ds_store"):
        return set()
    if mn.startswith("v_cmp") or mn.startswith("v_readlane") or mn.startswith("v_readfirstlane") or mn == "v_nop":
        return set()
    o = split_ops(ops)
    if not o:
        return set()
    w = vregs(o[0])
    if mn.startswith("v_swap") and len(o) > 1:
        w |= vregs(o[1])
    if ("lds" in o[1:]) and mn.startswith("buffer_load"):        # LDS-DMA: no VGPR destination
        return set()
    return w


def wait_states(mn, ops):
    if mn == "s_nop":
        return int(ops.split()[0], 0) + 1
    return 1


def branch_target(addr, mn, ops):
    if mn.startswith("s_cbranch") or mn == "s_branch":
        try:
            imm = int(ops.split()[0], 0)
        except ValueError:
            return None
        if imm >= 0x8000:
            imm -= 0x10000
        return addr + 4 + 4 * imm
    return None


def scan_kernel(name, ins):
    index = {a: i for i, (a, _, _) in enumerate(ins)}
    found = []
    for i, (addr, mn, ops) in enumerate(ins):
        if not STORE.match(mn):
            continue
        o = split_ops(ops)
        data = vregs(o[0]) if mn.startswith("buffer") else vregs(o[1])      # buffer: vdata first; global/flat: vaddr, vdata
        soffset_sgpr = mn.startswith("buffer") and len(o) >= 4 and re.match(r"^s\d+$", o[3].split()[0]) is not None
        best = None                                                          # (wait states, writer, async?)
        stack, seen = [(i + 1, 0)], set()
        while stack:
            j, ws = stack.pop()
            while j < len(ins) and ws < WINDOW and (j, ws) not in seen:
                seen.add((j, ws))
                a2, m2, o2 = ins[j]
                hit = written_vgprs(m2, o2) & data
                if hit:
                    cand = (ws, "%s %s" % (m2, o2), bool(ASYNC.match(m2)))
                    if best is None or (cand[2], cand[0]) < (best[2], best[0]):
                        best = cand
                    break
                if m2 == "s_endpgm":
                    break
                tgt = branch_target(a2, m2, o2)
                if tgt is not None and tgt in index:
                    stack.append((index[tgt], ws + 1))
                    if m2 == "s_branch":
                        break
                ws += wait_states(m2, o2)
                j += 1
        if best is not None:
            found.append({"kernel": name, "addr": "%x" % addr, "store": "%s %s" % (mn, ops), "sgpr_soffset": soffset_sgpr,
                          "wait_states": best[0], "writer": best[1], "async_writer": best[2]})
    return found


def scan(paths):
    rows, nstores = [], 0
    with tempfile.TemporaryDirectory() as tmp:
        for p in paths:
            for co in code_objects(p, tmp):
                for name, ins in disassemble(co).items():
                    nstores += sum(1 for _, mn, _ in ins if STORE.match(mn))
                    rows += scan_kernel(name, ins)
    return rows, nstores


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("paths", nargs="+")
    ap.add_argument("--min-wait", type=int, default=2)
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    rows, nstores = scan(a.paths)
    sync = [r for r in rows if not r["async_writer"]]
    bad = [r for r in sync if r["wait_states"] < a.min_wait]
    if a.json:
        print(json.dumps({"stores": nstores, "rewritten_within_window": len(sync), "below_min_wait": bad}, indent=1))
    else:
        print("%d stores of >= 96 bits; %d have their data VGPRs rewritten within %d wait states; %d below %d wait states" %
              (nstores, len(sync), WINDOW, len(bad), a.min_wait))
        hist = {}
        for r in sync:
            hist[r["wait_states"]] = hist.get(r["wait_states"], 0) + 1
        print("wait-state histogram (synchronous writers):", dict(sorted(hist.items())))
        for r in bad:
            print("  %s +%s: %s | %d wait state(s) | %s%s" % (r["kernel"][:70], r["addr"], r["store"], r["wait_states"], r["writer"],
                                                           " | SGPR soffset (LLVM: exempt)" if r["sgpr_soffset"] else ""))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
