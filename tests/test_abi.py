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


def test_result_record_is_56_bytes():
    import bluerov2_amd
    assert bluerov2_amd.RESULT_DTYPE.itemsize == 56


def test_default_options_match_reference_generated_solver():
    import bluerov2_amd
    o = bluerov2_amd.SolverOptions(80, 0.0125)
    # c_generated_code/acados_solver_bluerov2.c:422-481 (W), :559-566 (bounds), :668 (qp_iter_max)
    assert list(o.W) == [300, 480, 200, 10, 10, 200, 40, 40, 10, 10, 10, 10, 1, 1, 0.1, 0.05]
    assert list(o.We) == [300, 480, 200, 10, 10, 200, 40, 40, 10, 10, 10, 10]
    assert list(o.lbu) == [-50] * 4 and list(o.ubu) == [50] * 4 and o.qp_iter_max == 50


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
