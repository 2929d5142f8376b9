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
bounds).  The check walks every join block reached by an s_cbranch_execz that follows an s_*_saveexec (and / andn2 / or / xor
forms; up to four scalar instructions may sit between the two) and reports vector instructions -- v_accvgpr_read / _write copies
included -- between the join label and the first instruction that writes exec, in any of its shapes: `s_or_b64 exec, exec, saved`
(if-join), `s_andn2_saveexec_b64` / `s_or_saveexec_b64` / `s_xor_b64 exec, ...` (the else arm opening), `s_mov_b64 exec, saved`.  Input: shared libraries / objects with bundled code
objects of the target architecture, or assembly (.s) files written by `hipcc --cuda-device-only -S`.  Exit status 1 when anything
is found, 2 when the check could not run (no disassembler, no code object of the architecture): a build must not pass unchecked.

    check_exec_restore.py [--arch gfx950] [--objdump PATH] files...
The disassembler is looked for in: --objdump, $LLVM_OBJDUMP, next to $HIPCC, $ROCM_PATH/lib/llvm/bin, /opt/rocm/lib/llvm/bin, PATH.
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

import shutil

ARCH = "gfx950"
OBJDUMP = None


def find_objdump(explicit=None):
    cands = [explicit, os.environ.get("LLVM_OBJDUMP")]
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc")
    if hipcc:
        root = os.path.dirname(os.path.dirname(os.path.realpath(hipcc)))
        cands += [os.path.join(root, "lib", "llvm", "bin", "llvm-objdump"), os.path.join(root, "llvm", "bin", "llvm-objdump")]
    for env in ("ROCM_PATH", "ROCM_HOME"):
        if os.environ.get(env):
            cands.append(os.path.join(os.environ[env], "lib", "llvm", "bin", "llvm-objdump"))
    cands += ["/opt/rocm/lib/llvm/bin/llvm-objdump", shutil.which("llvm-objdump")]
    for c in cands:
        if c and os.path.isfile(c) and os.access(c, os.X_OK):
            return c
    return None


VECTOR = re.compile(r"^(v_|ds_|global_|buffer_|flat_|scratch_)")
LANE_INDEPENDENT = re.compile(r"^v_(readlane|writelane|readfirstlane)")   # not affected by exec


def code_objects(path):
    """the code objects of the target architecture in the clang offload bundles inside `path`"""
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
            if ARCH in ident and size:
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
    label_all = {}   # the disassembler numbers its labels per function (L43 exists many times): a branch means the nearest one
    for k, (_, t) in enumerate(ins):
        if t.startswith("LABEL "):
            label_all.setdefault(t[6:], []).append(k)
    func, hits = "?", []
    for k, (n, t) in enumerate(ins):
        if t.startswith("LABEL ") and not re.match(r"^LABEL (\.L|L\d+$)", t):
            func = t[6:]
        m = re.match(r"^s_cbranch_execz (\S+)", t)
        if not m or k == 0:
            continue
        # the mask saved by the nearest s_*_saveexec above the branch (scalar instructions that do not touch it may sit in between)
        ms = None
        for back in range(1, 6):
            if k - back < 0:
                break
            tb = ins[k - back][1]
            ms = re.match(r"^s_\w+_saveexec_b64 (s\[\d+:\d+\])", tb)
            if ms or tb.startswith("LABEL ") or not tb.startswith("s_") or re.match(r"^s_(cbranch|branch|endpgm|setpc)", tb):
                break
        tgt = min(label_all.get(m.group(1), []), key=lambda j: abs(j - k), default=None)
        if not ms or tgt is None:
            continue
        saved, bad = ms.group(1), []
        # Everything between the join label and the FIRST instruction that writes exec runs under the mask the branch arrived
        # with (one arm's lanes): the restore `s_or_b64 exec, exec, saved` / `s_mov_b64 exec, saved` of an if-join, or the
        # `s_andn2_saveexec` / `s_or_saveexec` / `s_xor_b64 exec` that opens the else arm.  Vector instructions in that gap are the
        # defect, whatever shape the exec write has.
        writes_exec = re.compile(r"^s_\w+ exec\b|^s_\w*saveexec_b64\b")
        for n2, t2 in ins[tgt + 1:]:
            if t2.startswith("LABEL ") or re.match(r"^s_(cbranch|branch|endpgm|setpc)", t2):
                break                                              # left the block without touching exec: not the pattern
            if writes_exec.match(t2):
                if bad:
                    hits.append((func, n, m.group(1), saved, bad))
                break
            if VECTOR.match(t2) and not LANE_INDEPENDENT.match(t2):
                bad.append((n2, t2))
    return hits


def main(argv):
    global ARCH, OBJDUMP
    paths, explicit = [], None
    it = iter(argv)
    for a in it:
        if a == "--arch":
            ARCH = next(it)
        elif a == "--objdump":
            explicit = next(it)
        else:
            paths.append(a)
    if any(not p.endswith(".s") for p in paths):
        OBJDUMP = find_objdump(explicit)
        if not OBJDUMP:
            print("check_exec_restore: no llvm-objdump found (tried --objdump, $LLVM_OBJDUMP, next to hipcc, $ROCM_PATH, /opt/rocm, PATH): "
                  "the device code was NOT checked", file=sys.stderr)
            return 2
    total = 0
    for p in paths:
        lines = listing(p)
        if not p.endswith(".s") and not any(ln.strip() for ln in lines):
            print(f"check_exec_restore: {p} holds no {ARCH} code object: nothing was checked", file=sys.stderr)
            return 2
        hits = scan(lines)
        total += len(hits)
        print(f"{os.path.basename(p)}: {len(hits)} join block(s) with vector instructions ahead of the exec restore")
        for func, line, lab, saved, bad in hits:
            print(f"  {func}: s_cbranch_execz at listing line {line} -> {lab} (mask saved in {saved}): {len(bad)} instruction(s) under "
                  f"the narrowed mask, first: line {bad[0][0]}: {bad[0][1]}")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
