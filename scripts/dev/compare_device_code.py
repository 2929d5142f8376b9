#!/usr/bin/env python3
"""compare_device_code.py A B -- are the gfx950 code objects bundled in two host objects / libraries the same code?
Prints, per kernel symbol, instruction count and a hash of its disassembly (addresses stripped) for both files and the verdict.
Used in round 5 to show that cutting qp_kernel.hip into qp/*.hpp changed no instruction (profiles/r5_split_isa.txt)."""
import hashlib
import re
import sys
import os

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import check_exec_restore as C


def kernels(path):
    C.OBJDUMP = C.find_objdump()
    out, name, body = {}, None, []
    for ln in C.listing(path):
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", ln)
        if m:
            if name and not name.startswith("L"):
                out[name] = body
            elif name:                       # a label inside the current function
                body_prev.append(ln.split("<")[1]); name, body = name_prev, body_prev
            name_prev, body_prev = name, body
            if not m.group(1).startswith("L"):
                name, body = m.group(1), []
                name_prev, body_prev = name, body
            continue
        t = ln.split("//")[0].strip()
        if name and t:
            body.append(t)
    if name:
        out[name] = body
    return out


a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
same = True
for k in sorted(set(a) | set(b)):
    ha = hashlib.md5("\n".join(a.get(k, [])).encode()).hexdigest()[:12]
    hb = hashlib.md5("\n".join(b.get(k, [])).encode()).hexdigest()[:12]
    ok = ha == hb
    same &= ok
    print(f"{k[:70]:70s} {len(a.get(k, [])):7d} {ha}   {len(b.get(k, [])):7d} {hb}   {'same' if ok else 'DIFFERENT'}")
print("IDENTICAL device code" if same else "device code DIFFERS")
sys.exit(0 if same else 1)
