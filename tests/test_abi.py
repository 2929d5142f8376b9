"""The C-ABI library loads, exports every symbol that include/bluerov2_nmpc.h declares, and refuses to run without a GPU
(no CPU fallback).  No compute is launched here."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(brov_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import bluerov2_amd
    bluerov2_amd.build_library()
    lib = ctypes.CDLL(bluerov2_amd.library_path())
    names = _declared("bluerov2_nmpc.h")
    assert len(names) > 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_result_record_layout():
    """104-byte record = brov_result of include/bluerov2_nmpc.h = orc_result: what the all-gather carries"""
    import bluerov2_amd
    from oracle import oracle_ffi
    d = bluerov2_amd.RESULT_DTYPE
    assert d.itemsize == 104 and d == oracle_ffi.RESULT_DTYPE
    assert [d.fields[k][1] for k in ("u0", "cost", "kkt", "status", "qp_iter", "thrust")] == [0, 32, 40, 48, 52, 56]
    from bluerov2_amd import distributed as D
    assert D.RECORD_BYTES == 104


def _defaults_fixture():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "solver_defaults.json")))


def test_default_options_match_reference_generator_dump(oracle):
    """brov_default_opts and orc_default_opts against the options the reference's generator dumped
    (bluerov2_dobmpc/scripts/acados_ocp.json -> tests/golden/solver_defaults.json by scripts/make_golden.py), not against
    hand-typed lists; the generated C renders the same values (acados_solver_bluerov2.c:422-481, :559-566, :668)."""
    import bluerov2_amd
    g = _defaults_fixture()
    assert (g["nx"], g["nu"], g["np"], g["ny"], g["ny_e"]) == (12, 4, 16, 16, 12) and g["idxbu"] == [0, 1, 2, 3]
    assert g["time_steps_uniform"] and g["x0_is_equality"] and abs(g["tf"] / g["N"] - g["time_step"]) < 1e-15
    assert (g["nlp_solver_type"], g["qp_solver"], g["hessian_approx"], g["integrator_type"], g["globalization"]) == (
        "SQP_RTI", "FULL_CONDENSING_HPIPM", "GAUSS_NEWTON", "ERK", "FIXED_STEP")
    assert g["sim_method_num_stages"] == 4 and g["sim_method_num_steps"] == 1 and g["nlp_solver_step_length"] == 1.0
    assert g["levenberg_marquardt"] == 0.0
    o = bluerov2_amd.SolverOptions(g["N"], g["time_step"])
    oo = oracle.opts(g["N"], g["time_step"])
    for opt, get in ((o, lambda k: list(getattr(o, k))), (oo, lambda k: list(getattr(oo, k)))):
        assert get("W") == g["W_diag"] and get("We") == g["We_diag"]
        assert get("lbu") == g["lbu"] and get("ubu") == g["ubu"]
        assert opt.qp_iter_max == g["qp_solver_iter_max"] and opt.N == g["N"] and opt.Ts == g["time_step"]
    # create-time defaults of the iterate / reference / parameters (checked on the oracle here, on the GPU in test_gpu_edge)
    x, u, pi, lam = oracle.init_iterate(oo)
    assert np.array_equal(x, np.tile(g["x0"], (g["N"] + 1, 1))) and not u.any() and not pi.any() and not lam.any()
    assert not any(g["yref"]) and not any(g["yref_e"]) and not any(g["parameter_values"])


def test_no_cpu_fallback():
    import torch
    import bluerov2_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(bluerov2_amd.NoDeviceError):
        bluerov2_amd.BatchSolver(4, bluerov2_amd.SolverOptions(20))


def test_ekf_has_no_cpu_fallback_and_reference_defaults():
    import torch
    import bluerov2_amd
    p = bluerov2_amd.EkfParams.default()
    # bluerov2_dob.h:171-183,208; bluerov2_dob.cpp:59-62
    assert p.dt == 0.05 and p.mass == 11.26 and p.bouyancy == 0.661618 and p.fd_step == 1e-6
    assert list(p.Dl) == [-11.7391, -20, -31.8678, -25, -44.9085, -5] and list(p.added_mass) == [1.7182, 0, 5.468, 0, 1.2481, 0.4006]
    assert p.R == 0.05 ** 4 / 4 and list(p.Q) == [0.05 ** 4 / 4] * 6 + [0.05 ** 2] * 12
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(bluerov2_amd.NoDeviceError):
        bluerov2_amd.BatchEkf(4)


def test_thrust_allocation_host_helper():
    import bluerov2_amd
    c = 0.026546960744430276
    t = bluerov2_amd.thrust_allocation(np.array([1.0, -2.0, 3.0, 0.5]))
    assert np.allclose(t, [(-1 - 2 + 0.5) / c, (-1 + 2 - 0.5) / c, (1 - 2 - 0.5) / c, (1 + 2 + 0.5) / c, -3 / c, -3 / c])


def test_group_api_links_rccl_and_refuses_without_a_gpu():
    """brov_group_* (several GPUs in one process): RCCL is resolved at run time -- the library loads without it being linked, the
    loader finds it, and without a GPU the create call refuses (no CPU fallback)"""
    import subprocess
    import torch
    import bluerov2_amd
    out = subprocess.run(["ldd", bluerov2_amd.library_path()], capture_output=True, text=True).stdout
    assert "rccl" not in out                                  # no link-time dependency
    assert bluerov2_amd.rccl_version() > 20000                # dlopen + ncclGetVersion
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(bluerov2_amd.NoDeviceError):
        bluerov2_amd.SolverGroup([0], 8, bluerov2_amd.SolverOptions(20))
