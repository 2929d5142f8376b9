#!/usr/bin/env python3
"""isa_loops.py <object or library> <kernel name substring> [min instructions] -- instruction mix of every LOOP of one gfx950 kernel (a loop = a
backward branch and everything between its target and itself; nested loops are listed inside-out, so the innermost entry of a nest is the body).
Round 6 (verdict item 3): the per-phase tables of rti_fused_kernel -- factor stage, forward, adjoint, roll-out, the trips of the linearisation --
are the rows of this listing (profiles/r6_fused_phase_isa.txt says which loop is which phase and how often it runs on the early-exit path)."""
import collections
import os
import re
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import check_exec_restore as C

CL = [("mfma", r"^v_mfma"), ("f64 dpp", r"^v_.*_f64_dpp"), ("f64", r"^v_(fma|mul|add|fmac|fmamk|fmaak|max|min|rcp|rsq|ldexp)_f64"),
      ("cmp", r"^v_cmp"), ("cndmask", r"^v_cndmask"), ("readlane", r"^v_read"), ("accvgpr", r"^v_accvgpr"),
      ("mov/bit", r"^v_(mov|writelane|bfi|and|or|xor|not|lshl|lshr|ashr|perm|swap|bfe)"),
      ("mul32", r"^v_(mul_lo|mul_hi|mad_u64)"), ("int", r"^v_(add|sub|mul|mad|lshl_add|add_lshl|min|max)_(u|i|co|nc)"), ("valu other", r"^v_"),
      ("lds", r"^ds_"), ("vmem", r"^(global|flat|buffer|scratch)_"), ("wait/nop", r"^s_(waitcnt|nop|barrier|sleep|setprio)"),
      ("branch", r"^s_(cbranch|branch|endpgm)"), ("salu", r"^s_")]
VALU = ("f64 dpp", "f64", "cmp", "cndmask", "readlane", "accvgpr", "mov/bit", "mul32", "int", "valu other")


def cls(m):
    for n, p in CL:
        if re.match(p, m):
            return n
    return "?"


def main():
    path, want = sys.argv[1], sys.argv[2]
    minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    C.OBJDUMP = C.find_objdump()
    name, rows = None, collections.OrderedDict()
    for ln in C.listing(path):
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", ln)
        if m and not m.group(1).startswith("L"):
            name = m.group(1)
        if name and want in name:
            rows.setdefault(name, []).append(ln)
    for k, lines in rows.items():
        lab = {}
        for i, l in enumerate(lines):
            m = re.match(r"^[0-9a-f]+ <(L\d+)>:", l)
            if m:
                lab[m.group(1)] = i
        ins = [None] * len(lines)
        for i, l in enumerate(lines):
            t = l.split("//")[0].split()
            if t and not t[0].endswith(":") and not re.match(r"^[0-9a-f]+$", t[0]):
                ins[i] = t
        total = collections.Counter(cls(t[0]) for t in ins if t)
        print(f"{k}: {sum(total.values())} instructions; vector port: {sum(v for c, v in total.items() if c in VALU)} VALU + {total['mfma']} MFMA")
        seen = set()
        for i, t in enumerate(ins):
            if t and (t[0].startswith("s_cbranch") or t[0] == "s_branch") and t[-1] in lab and lab[t[-1]] < i:
                a = lab[t[-1]]
                h = collections.Counter(cls(x[0]) for x in ins[a:i + 1] if x)
                n = sum(h.values())
                if n < minlen or (a, n) in seen:
                    continue
                seen.add((a, n))
                valu = sum(v for c, v in h.items() if c in VALU)
                print(f"  loop at +{a:5d} .. +{i:5d}: {n:5d} instr, {valu:5d} VALU, {h['mfma']:3d} MFMA | " +
                      " ".join(f"{c}={v}" for c, v in sorted(h.items(), key=lambda x: -x[1]) if c != "mfma"))


main()
