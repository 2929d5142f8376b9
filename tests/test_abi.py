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


def test_thrust_allocation_host_helper():
    import bluerov2_amd
    c = 0.026546960744430276
    t = bluerov2_amd.thrust_allocation(np.array([1.0, -2.0, 3.0, 0.5]))
    assert np.allclose(t, [(-1 - 2 + 0.5) / c, (-1 + 2 - 0.5) / c, (1 - 2 - 0.5) / c, (1 + 2 + 0.5) / c, -3 / c, -3 / c])
