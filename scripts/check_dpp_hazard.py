#!/usr/bin/env python3
"""check_dpp_hazard.py <object or library> [kernel name substring] -- static check of the gfx950 code for the one data hazard the hardware
does not interlock and the compiler cannot see through inline assembly: a VALU instruction writes a VGPR and a DPP instruction reads
that VGPR as its DPP operand (src0) less than two wait states later (ISA guide, "VALU writes VGPR -> DPP reads that VGPR: 2 wait
states").  The EKF kernels issue their row-broadcast multiply-adds as `asm volatile("v_fmac_f64_dpp ...")`; the compiler schedules the
instructions that PRODUCE the broadcast rows around those statements freely (round 6: a changed operand form moved the finite-
difference arithmetic of F^T into the first product, which then read stale registers).  The sources pin the rows ahead of every product
(pin_rows in ekf_kernel.hip); this script is the proof on the code that ships.  Exit status 1 and a listing if a hazard is found.

A wait state is one issued instruction or one count of s_nop (s_nop N = N + 1).  Branch targets are handled conservatively: the scan is
over the straight-line listing, and a label does not reset the window (a jump INTO the window from elsewhere is not modelled; the
products are straight-line code)."""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_exec_restore as C


def regs(tok):
    """set of VGPR numbers named by an operand token like v12, v[12:13], -v[4:5], |v3|"""
    m = re.search(r"\bv\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.search(r"\bv(\d+)\b", tok)
    return {int(m.group(1))} if m else set()


def scan(lines):
    hits, name = [], "?"
    window = []   # [(wait states this instruction contributes, set of VGPRs it writes, text)] most recent last
    for ln in lines:
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", ln)
        if m:
            if not m.group(1).startswith("L"):
                name, window = m.group(1), []
            continue
        t = ln.split("//")[0].strip()
        t = re.sub(r"^[0-9a-f]+\s+", "", t) if re.match(r"^[0-9a-f]{8,}\s", t) else t
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        ops = [o.strip() for o in t[len(op):].split(",")]
        if "row_" in t or "quad_perm" in t or "wave_" in t or op.endswith("_dpp"):
            # DPP operand = src0 = the operand behind the destination
            src0 = regs(ops[1]) if len(ops) > 1 else set()
            ws = 0
            for w, wr, txt in reversed(window):
                if ws >= 2:
                    break
                if wr & src0:
                    hits.append((name, txt, t, ws))
                    break
                ws += w
        written = set()
        if op.startswith("v_") and not op.startswith("v_cmp") and not op.startswith("v_readlane") and not op.startswith("v_readfirstlane"):
            written = regs(ops[0]) if ops else set()
        w = 1
        m = re.match(r"s_nop\s+(\d+)", t)
        if m:
            w = int(m.group(1)) + 1
        window.append((w, written, t))
        if len(window) > 4:
            window.pop(0)
    return hits


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    C.OBJDUMP = C.find_objdump()
    hits = [h for h in scan(C.listing(path)) if want in h[0]]
    for name, wr, rd, ws in hits:
        print(f"{name}: `{wr}` is followed {ws} wait state(s) later by `{rd}`")
    print(f"{os.path.basename(path)}: {len(hits)} VALU-write -> DPP-read hazard(s)")
    return 1 if hits else 0


if __name__ == "__main__":
    sys.exit(main())
