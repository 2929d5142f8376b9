#!/usr/bin/env python3
"""isa_histogram.py <object or library> <kernel name substring> -- static instruction mix of one gfx950 kernel, by issue class.
Round 5 (verdict item 4): the per-instruction table of the linearisation.  lin_wave_kernel IS lin_phase<TWO> + the staging in front and
the copy-out behind it, and at N = 20 (three lanes per interval) each of its `#pragma unroll 1` column loops runs exactly one trip, so
the static count of that kernel is the dynamic count of the phase (profiles/r5_lin_phase_isa.txt)."""
import collections
import os
import re
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import check_exec_restore as C

CLASSES = [
    ("mfma", r"^v_mfma"),
    ("valu f64 fma/mul/add", r"^v_(fma|mul|add|fmac|fmamk|fmaak)_f64"),
    ("valu f64 other (max/min/rcp/rsq/ldexp/frexp/trig_preop/floor/fract/cvt)", r"^v_.*_f64|^v_cvt_.*f64"),
    ("valu compare / select", r"^v_cmp|^v_cndmask|^v_cmpx"),
    ("valu move / dpp / bit (32-bit)", r"^v_(mov|accvgpr|readlane|readfirstlane|writelane|bfi|and|or|xor|not|lshl|lshr|ashr|perm|swap|bfe)"),
    ("valu integer arithmetic", r"^v_(add|sub|mul|mad|lshl_add|add_lshl|min|max)_(u|i|co|nc)"),
    ("valu other", r"^v_"),
    ("lds", r"^ds_"),
    ("vmem", r"^(global|flat|buffer|scratch)_"),
    ("smem", r"^s_(load|buffer_load|memtime|memrealtime|dcache)"),
    ("waitcnt / nop / barrier", r"^s_(waitcnt|nop|barrier|sleep|setprio)"),
    ("branch", r"^s_(cbranch|branch|endpgm|setpc|swappc|call)"),
    ("salu", r"^s_"),
]


def main():
    path, want = sys.argv[1], sys.argv[2]
    C.OBJDUMP = C.find_objdump()
    name, rows = None, collections.OrderedDict()
    for ln in C.listing(path):
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", ln)
        if m and not m.group(1).startswith("L"):
            name = m.group(1)
            continue
        t = ln.split("//")[0].strip()
        if name and want in name and t and not t.endswith(":") and not re.match(r"^<", t):
            rows.setdefault(name, []).append(t.split()[0])
    for k, ins in rows.items():
        hist = collections.Counter()
        mn = collections.Counter(ins)
        for i in ins:
            for cname, pat in CLASSES:
                if re.match(pat, i):
                    hist[cname] += 1
                    break
            else:
                hist["?"] += 1
        total = len(ins)
        valu = sum(v for c, v in hist.items() if c.startswith("valu") or c == "mfma")
        print(f"{k}: {total} instructions, {valu} on the vector issue port (x 4 cycles = {4 * valu})")
        for cname, _ in CLASSES + [("?", "")]:
            if hist[cname]:
                print(f"  {cname:75s} {hist[cname]:6d}  {100.0 * hist[cname] / total:5.1f} %")
        print("  most frequent mnemonics: " + ", ".join(f"{m} {n}" for m, n in mn.most_common(24)))


main()
