"""The solver kernels must not use scratch memory.  Round 2 met a build of the windowed kernel whose register spills went to
scratch and whose linearisation then produced wrong sensitivity columns (values kept live to the end of lin_phase came back as
zeros; the same source without spills was correct).  The backend's own resource report (hipcc -Rpass-analysis=kernel-resource-
usage, device-only compile of qp_kernel.hip, ~12 s) is checked here so that a change that pushes a kernel back into scratch
fails on the CPU, before any GPU test runs."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _report(src):
    out = subprocess.run(["bash", os.path.join(ROOT, "scripts", "dev", "kernel_resources.sh"), src], capture_output=True, text=True,
                         timeout=600).stdout
    rep = {}
    for ln in out.splitlines():
        m = re.match(r"Name: (\S+)", ln)
        if not m:
            continue
        name = m.group(1)
        rep[name] = {k: int(v) for k, v in re.findall(r"\|([A-Za-z ]+): (\d+)", ln)}
    return rep


def test_solver_kernels_use_no_scratch_and_keep_their_occupancy():
    rep = _report("qp_kernel.hip")
    want = {"rti_fused_kernel": 1, "rti_fused_kernel_w2": 2, "rti_fused_kernel_grid": 1, "rti_window_kernel": 1, "rti_window_kernel_grid": 1,
            "rti_window_kernel_res": 1, "rti_window_kernel_res_grid": 1, "rti_window_kernel_res_split": 1, "rti_window_kernel_res_split_grid": 1, "rti_pit_kernel": 1, "rti_pit_kernel_grid": 1, "rti_pit_kernel_fb": 1, "rti_pit_kernel_fb_grid": 1, "rti_fused_kernel_mail": 1, "qp_kernel": 2, "lin_wave_kernel": 1, "lin_wave_kernel_grid": 1}
    seen = {}
    for mangled, r in rep.items():
        for short in want:
            if re.search(r"\d+%sE" % short, mangled):   # Itanium mangling: <length><name>E
                seen[short] = r
    assert set(seen) == set(want), (sorted(seen), sorted(rep))
    for short, occ in want.items():
        r = seen[short]
        assert r["scratch"] == 0, (short, r)
        assert r["occ"] == occ, (short, r)
        assert r["VGPRs"] <= 256, (short, r)


def test_ekf_kernel_keeps_two_waves_per_simd_and_seven_per_cu():
    """the structured EKF kernel's rate (177 M updates/s against 117 M of the dense one) is its occupancy: at most 256 registers
    (two waves per SIMD) without scratch, and two 18 x 18 LDS buffers per filter (22.5 KB per wave: seven waves per CU of 160 KB)"""
    rep = _report("ekf_kernel.hip")
    sp = [r for m, r in rep.items() if re.search(r"\d+ekf_update_kernel_spE", m)]
    assert len(sp) == 1, sorted(rep)
    assert sp[0]["scratch"] == 0 and sp[0]["occ"] >= 2 and sp[0]["VGPRs"] <= 256, sp[0]
    src = open(os.path.join(ROOT, "bluerov2_amd", "csrc", "ekf_kernel.hip")).read()
    m = re.search(r"constexpr int kSpLds = (\d+) \* kMat \+ (\d+) \* EN;", src)
    assert m, "kSpLds"
    per_wave = 4 * (int(m.group(1)) * 324 + int(m.group(2)) * 18) * 8
    assert 160 * 1024 // per_wave >= 7, per_wave


# ---- code-generation defect found in round 3 (scripts/check_exec_restore.py): copies of live registers ahead of a join block's
# exec restore.  The Makefile runs the check on the library it links; here: the checker itself, and the library in the tree.
_BAD = """
_ZN4brov6kernelE:
	s_and_saveexec_b64 s[4:5], s[6:7]
	s_cbranch_execz .LBB5_323
; %bb.320:
	global_atomic_add v3, v3, v4, s[6:7] sc0
.LBB5_323:
	v_accvgpr_write_b32 a64, v162
	v_writelane_b32 v255, s4, 3
	v_accvgpr_write_b32 a65, v163
	s_or_b64 exec, exec, s[4:5]
	s_endpgm
"""
_GOOD = _BAD.replace("\tv_accvgpr_write_b32 a64, v162\n\tv_writelane_b32 v255, s4, 3\n\tv_accvgpr_write_b32 a65, v163\n\ts_or_b64 exec, exec, s[4:5]\n",
                     "\tv_writelane_b32 v255, s4, 3\n\ts_or_b64 exec, exec, s[4:5]\n\tv_accvgpr_write_b32 a64, v162\n\tv_accvgpr_write_b32 a65, v163\n")


def _checker():
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_exec_restore", os.path.join(ROOT, "scripts", "check_exec_restore.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_exec_restore_checker_recognises_the_defect():
    chk = _checker()
    hits = chk.scan(_BAD.split("\n"))
    assert len(hits) == 1 and hits[0][0] == "_ZN4brov6kernelE" and hits[0][3] == "s[4:5]"
    assert [t for _, t in hits[0][4]] == ["v_accvgpr_write_b32 a64, v162", "v_accvgpr_write_b32 a65, v163"]   # v_writelane ignores exec
    assert chk.scan(_GOOD.split("\n")) == []


# round 4: the else-join shapes (the mask is transformed by s_xor between the saveexec and the branch, and the join block opens the
# else arm with s_andn2_saveexec / s_or_saveexec instead of restoring), AGPR reads, and the per-function label numbering of the
# disassembler (L43 exists once per kernel: a branch means the nearest one)
_BAD_ELSE = """
_ZN4brov6kernelE:
	s_and_saveexec_b64 s[8:9], vcc
	s_xor_b64 s[8:9], exec, s[8:9]
	s_cbranch_execz L4
	v_mov_b32 v1, v2
L4:
	v_accvgpr_read_b32 v3, a1
	s_andn2_saveexec_b64 s[8:9], s[8:9]
	v_mov_b32 v1, v2
	s_or_b64 exec, exec, s[8:9]
	s_endpgm
_ZN4brov6otherE:
	s_and_saveexec_b64 s[4:5], s[6:7]
	s_cbranch_execz L4
	v_mov_b32 v1, v2
L4:
	s_or_b64 exec, exec, s[4:5]
	v_accvgpr_write_b32 a64, v162
	s_endpgm
"""


def test_exec_restore_checker_knows_the_else_join_and_per_function_labels():
    chk = _checker()
    hits = chk.scan(_BAD_ELSE.split("\n"))
    assert len(hits) == 1 and hits[0][0] == "_ZN4brov6kernelE" and [t for _, t in hits[0][4]] == ["v_accvgpr_read_b32 v3, a1"]
    good = _BAD_ELSE.replace("\tv_accvgpr_read_b32 v3, a1\n\ts_andn2_saveexec_b64 s[8:9], s[8:9]\n", "\ts_andn2_saveexec_b64 s[8:9], s[8:9]\n\tv_accvgpr_read_b32 v3, a1\n")
    assert chk.scan(good.split("\n")) == []


def test_exec_restore_checker_refuses_to_pass_unchecked(tmp_path):
    """a library without a code object of the architecture asked for, or no disassembler: exit status 2, not a silent pass"""
    import subprocess, sys
    lib = os.path.join(ROOT, "bluerov2_amd", "lib", "libbluerov2_nmpc.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    script = os.path.join(ROOT, "scripts", "check_exec_restore.py")
    assert subprocess.run([sys.executable, script, "--arch", "gfx90a", lib], capture_output=True).returncode == 2
    r = subprocess.run([sys.executable, script, "--objdump", str(tmp_path / "nope"), lib], capture_output=True,
                       env={"PATH": str(tmp_path), "HIPCC": str(tmp_path / "hipcc"), "ROCM_PATH": str(tmp_path)})
    assert r.returncode in (0, 2)   # 2 unless /opt/rocm provides the tool (the last resort the search still tries)


def test_shipped_library_has_no_vector_code_ahead_of_an_exec_restore():
    lib = os.path.join(ROOT, "bluerov2_amd", "lib", "libbluerov2_nmpc.so")
    chk = _checker()
    chk.OBJDUMP = chk.find_objdump()
    if not os.path.exists(lib) or not chk.OBJDUMP:
        pytest.skip("library not built / no llvm-objdump")
    lines = chk.listing(lib)
    assert sum("s_cbranch_execz" in ln for ln in lines) > 500      # the disassembly is there and symbolised
    assert chk.scan(lines) == []


def test_build_gate_catches_the_defect_in_real_compiler_output(tmp_path):
    """The checker against REAL compiler output.  (1) A committed excerpt of hipcc 7.2's own assembly of this repository's qp_kernel.hip
    (commit da0d7bb with -DBROV_SCHED_TICKET_LATE: the work-ordering ticket taken at the end of the wave, a harmless reordering) in which
    the register allocator's AGPR copies sit ahead of an exec restore in rti_fused_kernel: the checker must find it.  With BROV_TEST_CANARY=1 also: (2) the product
    order compiled now: nothing.  (3) The canary order compiled now: reported, not asserted -- which statement order builds the defect
    moves with every change of the source (rti_window_kernel, then rti_fused_kernel, then -- with the parallel-in-time kernel in the
    file -- nowhere), which is the reason the link rule runs the checker on every build."""
    chk = _checker()
    fixture = os.path.join(ROOT, "tests", "golden", "exec_restore_defect_hipcc72.s")
    hits = chk.scan(open(fixture).read().split("\n"))
    assert len(hits) == 1 and hits[0][0].endswith("rti_fused_kernelENS_9DevParamsE") and hits[0][2] == ".LBB3_47", hits
    assert all("v_accvgpr_write_b32" in ins for _, ins in hits[0][4])
    # (2), (3): two more device compiles of qp_kernel.hip (~75 s each).  The product order is already checked on the linked library
    # (Makefile link rule, test_shipped_library_has_no_vector_code_ahead_of_an_exec_restore); the live canary only prints.  On request:
    if os.environ.get("BROV_TEST_CANARY", "0") == "0":
        return
    src = os.path.join(ROOT, "bluerov2_amd", "csrc", "qp_kernel.hip")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    out = {}
    for name, extra in (("product", []), ("late", ["-DBROV_SCHED_TICKET_LATE"])):
        asm = tmp_path / f"qp_{name}.s"
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
                        "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", src, "-o", str(asm)] + extra,
                       check=True, capture_output=True, timeout=900)
        out[name] = chk.scan(open(asm).read().split("\n"))
    assert out["product"] == []
    print(f"canary order (-DBROV_SCHED_TICKET_LATE) compiled now: {len(out['late'])} join block(s) with the defect", [h[0] for h in out["late"]])


def test_device_integer_helpers_and_the_slow_multiplies_they_replace():
    """Round 6.  (1) qp/tiles.hpp div12(): j / 12 as (j * 43691) >> 19 -- one full-rate 24-bit multiply -- must be exact over the element
    indices the kernels form (up to 257 x 12 at N = 256) and its operands must fit 24 bits.  (2) v_mul_lo_u32 is a quarter-rate instruction on
    gfx950 (16 cycles); `stage index x per-lane stride` compiled to it two to four times per stage of every sweep until lmul() (v_mul_i32_i24)
    took over: the fused kernel's hot loops -- factor stage, forward, adjoint -- must stay free of the 32-bit multiplies."""
    import numpy as np
    j = np.arange(0, 131072, dtype=np.int64)
    assert np.array_equal((j * 43691) >> 19, j // 12) and 43691 < 2 ** 24 and j.max() < 2 ** 24
    assert ((np.int64(131075) * 43691) >> 19) != 131075 // 12   # (the bound in the comment is the real one)
    obj = os.path.join(ROOT, "bluerov2_amd", "lib", "obj", "qp_kernel.o")
    if not os.path.exists(obj):
        pytest.skip("no object file (library not built here)")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "dev", "isa_loops.py"), obj, "16rti_fused_kernelENS", "100"],
                         capture_output=True, text=True, check=True).stdout
    # three stages of a factor sweep / of a DPP sweep per trip (the stage loops themselves: under 800 instructions, not the QP loop around them)
    hot = [ln for ln in out.splitlines() if (" 30 MFMA" in ln or "f64 dpp=48" in ln) and int(re.search(r"(\d+) instr", ln).group(1)) < 800]
    assert len(hot) >= 3, out[:400]
    assert not any("mul32=" in ln for ln in hot), [ln for ln in hot if "mul32=" in ln]


# ---- data hazard found in round 6 (scripts/check_dpp_hazard.py): a DPP read of a VGPR less than two wait states behind the VALU write of it.
_HAZ = """
0000000000001000 <_ZN4brov6kernelE>:
	v_mul_f64 v[90:91], v[148:149], v[90:91]
	v_fmac_f64_dpp v[218:219], v[90:91], v[102:103] row_newbcast:13 row_mask:0xf bank_mask:0xf
	s_endpgm
"""
_HAZ1 = _HAZ.replace("\tv_fmac_f64_dpp", "\tv_add_f64 v[2:3], v[4:5], v[6:7]\n\tv_fmac_f64_dpp")
_OK_NOP = _HAZ.replace("\tv_fmac_f64_dpp", "\ts_nop 1\n\tv_fmac_f64_dpp")
_OK_TWO = _HAZ1.replace("\tv_fmac_f64_dpp", "\tds_read_b64 v[8:9], v10\n\tv_fmac_f64_dpp")
_OK_ACC = _HAZ.replace("v_mul_f64 v[90:91], v[148:149], v[90:91]", "v_mul_f64 v[218:219], v[148:149], v[90:91]")   # the accumulator is not the DPP operand


def _hazard_checker():
    import importlib.util
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    spec = importlib.util.spec_from_file_location("check_dpp_hazard", os.path.join(ROOT, "scripts", "check_dpp_hazard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_dpp_hazard_checker_recognises_the_hazard():
    chk = _hazard_checker()
    h = chk.scan(_HAZ.split("\n"))
    assert len(h) == 1 and h[0][0] == "_ZN4brov6kernelE" and h[0][3] == 0, h
    h = chk.scan(_HAZ1.split("\n"))
    assert len(h) == 1 and h[0][3] == 1, h
    for ok in (_OK_NOP, _OK_TWO, _OK_ACC):
        assert chk.scan(ok.split("\n")) == [], ok


def test_library_in_tree_has_no_dpp_hazard():
    lib = os.path.join(ROOT, "bluerov2_amd", "lib", "libbluerov2_nmpc.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    chk = _hazard_checker()
    chk.C.OBJDUMP = chk.C.find_objdump()
    assert chk.scan(chk.C.listing(lib)) == []
