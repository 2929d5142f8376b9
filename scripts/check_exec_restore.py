#!/usr/bin/env python3
"""Build-time check of the shipped device code for one specific code-generation defect (found in round 3):

    s_and_saveexec_b64 s[4:5], s[6:7]      ; if (lane == 0) ...
    s_cbranch_execz    JOIN
    ...
  JOIN:
    v_accvgpr_write_b32 a64, v162          ; register-allocator copies of LIVE registers, placed at the top of the join block
    ...                                    ; -- they run with ONE lane enabled: the other 63 lanes of the parked registers go stale
    s_or_b64 exec, exec, s[4:5]            ; the mask is restored only here

A harmless reordering of two source statements made hipcc (ROCm 7.2) emit this in rti_window_kernel; every pointer the kernel rebuilt
from the parked registers was then valid in lane 0 only (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION on the first QP with active
bounds).  The check walks every join block reached by an s_cbranch_execz that directly follows an s_*_saveexec and reports vector
instructions ahead of the matching exec restore.  Input: shared libraries / objects with bundled gfx950 code objects, or assembly
(.s) files written by `hipcc --cuda-device-only -S`.  Exit status 1 when anything is found.
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
VECTOR = re.compile(r"^(v_|ds_|global_|buffer_|flat_|scratch_)")
LANE_INDEPENDENT = re.compile(r"^v_(readlane|writelane|readfirstlane)")   # not affected by exec


def code_objects(path):
    """the gfx950 code objects of a clang offload bundle inside `path`"""
    data = open(path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, pos = [], 0
    while True:
        i = data.find(magic, pos)
        if i < 0:
            return out
        (cnt,) = struct.unpack_from("<Q", data, i + 24)
        p = i + 32
        for _ in range(cnt):
            off, size, idl = struct.unpack_from("<QQQ", data, p)
            p += 24
            ident = data[p:p + idl].decode(errors="replace")
            p += idl
            if "gfx950" in ident and size:
                out.append(data[i + off:i + off + size])
        pos = i + 24


def listing(path):
    """(text lines) of a .s file, or of the disassembly of every code object bundled in a library / object"""
    if path.endswith(".s"):
        return open(path).read().split("\n")
    lines = []
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            lines += subprocess.run([OBJDUMP, "-d", "--symbolize-operands", f.name], check=True, stdout=subprocess.PIPE,
                                    universal_newlines=True).stdout.split("\n")
    return lines


def scan(lines):
    ins = []      # (line number, text) of instructions / labels, comments stripped
    for n, ln in enumerate(lines, 1):
        t = ln.split("//")[0].split(";")[0].strip()
        m = re.match(r"^(?:[0-9a-f]+ )?<(\w+)>:$", t) or re.match(r"^([.\w$]+):", t)
        if m:
            ins.append((n, "LABEL " + m.group(1)))
        elif t and not t.startswith("."):
            ins.append((n, t))
    label_at = {t[6:]: k for k, (_, t) in enumerate(ins) if t.startswith("LABEL ")}
    func, hits = "?", []
    for k, (n, t) in enumerate(ins):
        if t.startswith("LABEL ") and not re.match(r"^LABEL (\.L|L\d+$)", t):
            func = t[6:]
        m = re.match(r"^s_cbranch_execz (\S+)", t)
        if not m or k == 0:
            continue
        ms = re.match(r"^s_\w+_saveexec_b64 (s\[\d+:\d+\])", ins[k - 1][1])
        tgt = label_at.get(m.group(1))
        if not ms or tgt is None:
            continue
        saved, bad = ms.group(1), []
        for n2, t2 in ins[tgt + 1:]:
            if t2.startswith("LABEL ") or re.match(r"^s_(cbranch|branch|endpgm|setpc)", t2):
                break                                              # left the block without touching exec: not the pattern
            if re.match(r"^s_or_b64 exec, exec, " + re.escape(saved) + r"$", t2):
                if bad:
                    hits.append((func, n, m.group(1), saved, bad))
                break
            if re.search(r"\bexec\b", t2):
                break                                              # some other exec manipulation: not the pattern
            if VECTOR.match(t2) and not LANE_INDEPENDENT.match(t2):
                bad.append((n2, t2))
    return hits


def main(paths):
    total = 0
    for p in paths:
        hits = scan(listing(p))
        total += len(hits)
        print(f"{os.path.basename(p)}: {len(hits)} join block(s) with vector instructions ahead of the exec restore")
        for func, line, lab, saved, bad in hits:
            print(f"  {func}: s_cbranch_execz at listing line {line} -> {lab} (mask saved in {saved}): {len(bad)} instruction(s) under "
                  f"the narrowed mask, first: line {bad[0][0]}: {bad[0][1]}")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
