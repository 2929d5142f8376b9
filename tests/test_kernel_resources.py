"""The solver kernels must not use scratch memory.  Round 2 met a build of the windowed kernel whose register spills went to
scratch and whose linearisation then produced wrong sensitivity columns (values kept live to the end of lin_phase came back as
zeros; the same source without spills was correct).  The backend's own resource report (hipcc -Rpass-analysis=kernel-resource-
usage, device-only compile of qp_kernel.hip, ~12 s) is checked here so that a change that pushes a kernel back into scratch
fails on the CPU, before any GPU test runs."""
import os
import re
import subprocess

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
    want = {"rti_fused_kernel": 1, "rti_fused_kernel_w2": 2, "rti_window_kernel": 1, "rti_window_kernel_res": 1, "qp_kernel": 2,
            "lin_wave_kernel": 1, "lin_wave_kernel_grid": 1}
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
