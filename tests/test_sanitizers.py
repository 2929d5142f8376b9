"""Host code under AddressSanitizer + UndefinedBehaviorSanitizer ("CPU restatement under ASan/UBSan"; VERDICT
round 5 item 6), on a CPU:

* the C oracle's single-step path, its OpenMP batch driver and its QP loop with active bounds (tests/san_batch_driver.c), against the
  optimised oracle library the rest of the suite uses;
* the acados-shaped drop-in's host logic -- bluerov2_amd/csrc/acados_shim.cpp, every line of it -- driven by tests/shim_caller.c through
  the reference's per-tick call sequence, with the fourteen brov_* calls it makes answered by a test double on top of the oracle
  (tests/brov_oracle_double.c): the known-answer ticks at N = 80, the preparation / feedback split, a non-uniform grid, a failed step
  (NaN measurement) under both failure policies, and the QP-iteration-limit tick.

Any sanitizer report aborts the binary (-fno-sanitize-recover=all) and fails the test; the answers are checked as well, so this is also
the CPU-side test of the shim's status mapping."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1", "-fopenmp"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
           OMP_NUM_THREADS="4")
ORACLE_C = [os.path.join(ROOT, "oracle", "bluerov2_oracle.c"), os.path.join(ROOT, "oracle", "bluerov2_ekf_oracle.c")]


@pytest.fixture(scope="module")
def san_bins(tmp_path_factory):
    d = tmp_path_factory.mktemp("san")
    objs = []
    for src in ORACLE_C + [os.path.join(ROOT, "tests", "brov_oracle_double.c"), os.path.join(ROOT, "tests", "shim_caller.c"),
                           os.path.join(ROOT, "tests", "san_batch_driver.c")]:
        o = str(d / (os.path.basename(src) + ".o"))
        subprocess.check_call(["gcc", "-std=gnu99", "-Wall", *SAN, f"-I{ROOT}/include", f"-I{ROOT}/include/acados_shim", "-c", src, "-o", o])
        objs.append(o)
    shim_o = str(d / "acados_shim.o")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", *SAN, f"-I{ROOT}/include", f"-I{ROOT}/include/acados_shim", "-c",
                           os.path.join(ROOT, "bluerov2_amd", "csrc", "acados_shim.cpp"), "-o", shim_o])
    caller, batch = str(d / "shim_caller_san"), str(d / "batch_san")
    subprocess.check_call(["g++", *SAN, "-o", caller, objs[0], objs[1], objs[2], objs[3], shim_o, "-lm"])
    subprocess.check_call(["gcc", *SAN, "-o", batch, objs[0], objs[1], objs[4], "-lm"])
    return caller, batch


def _run(exe, *args, env=None):
    r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=600, env=env or ENV)
    assert r.returncode == 0 and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr, \
        (r.returncode, r.stdout[-1500:], r.stderr[-4000:])
    return r.stdout


def _blob(tmp_path, g, name, nt):
    blob = np.concatenate([g[f"{name}/x0_meas"], g[f"{name}/p"][0], [float(nt)]] + [g[f"{name}/yref{k}"].ravel() for k in range(nt)])
    inp = tmp_path / "in.bin"
    inp.write_bytes(blob.astype(np.float64).tobytes())
    return str(inp)


@pytest.mark.parametrize("mode", ["", "S"])
def test_shim_known_answer_ticks_under_sanitizers(san_bins, tmp_path, golden_rti, mode):
    g, name, nt = golden_rti, "circle_N80", 4
    out = _run(san_bins[0], _blob(tmp_path, g, name, nt), *([mode] if mode else []))
    ticks = [ln.split() for ln in out.splitlines() if ln.startswith("TICK")]
    assert len(ticks) == nt
    for k, t in enumerate(ticks):
        assert int(t[3]) == 0
        assert np.abs(np.array([float(v) for v in t[9:13]]) - g[f"{name}/u{k}"][0]).max() < 1e-6
        assert np.abs(np.array([float(v) for v in t[14:17]]) - g[f"{name}/x{k}"][1, :3]).max() < 1e-6
    assert "custom_update 1" in out and "free 0" in out


def test_shim_non_uniform_grid_under_sanitizers(san_bins, tmp_path, golden_rti, oracle):
    g, name, nt = golden_rti, "circle_N80", 2
    out = _run(san_bins[0], _blob(tmp_path, g, name, nt), "G")
    ticks = [ln.split() for ln in out.splitlines() if ln.startswith("TICK")]
    ts, t = np.zeros(80), 0.008
    for i in range(80):
        ts[i] = t; t *= 1.01
    op = oracle.opts(80, 0.0125, ts_vec=ts)
    x, u = g[f"{name}/x_init"].copy(), g[f"{name}/u_init"].copy()
    pi, lam = np.zeros((80, 12)), np.zeros((80, 8))
    for k, tk in enumerate(ticks):
        oracle.rti_step(op, g[f"{name}/x0_meas"], g[f"{name}/yref{k}"], g[f"{name}/p"], x, u, pi, lam)
        assert int(tk[3]) == 0 and np.abs(np.array([float(v) for v in tk[9:13]]) - u[0]).max() < 1e-8


@pytest.mark.parametrize("policy", ["keep", "restart"])
def test_shim_failed_step_under_sanitizers(san_bins, tmp_path, golden_rti, policy):
    g, name = golden_rti, "circle_N80"
    env = dict(ENV)
    env.pop("BROV_ON_FAILURE", None)
    if policy == "restart":
        env["BROV_ON_FAILURE"] = "restart"
    out = _run(san_bins[0], _blob(tmp_path, g, name, 2), "F", env=env)
    last_good = np.array([float(v) for v in [ln for ln in out.splitlines() if ln.startswith("TICK")][-1].split()[9:13]])
    failed = [ln for ln in out.splitlines() if ln.startswith("FAILED")][0].split()
    assert int(failed[2]) == 1 and np.array_equal(np.array([float(v) for v in failed[4:8]]), last_good)
    rec = [ln for ln in out.splitlines() if ln.startswith("RECOVERED")][0].split()
    assert int(rec[2]) == 0 and np.all(np.isfinite([float(v) for v in rec[4:8]]))


@pytest.mark.parametrize("strict", [False, True])
def test_shim_iteration_limit_status_mapping_under_sanitizers(san_bins, tmp_path, golden_rti, strict):
    """the CPU-side twin of tests/test_gpu_shim.py::test_qp_iteration_limit_returns_success_like_sqp_rti: 0 from the call (acados' SQP_RTI
    returns ACADOS_SUCCESS when the QP stops at its iteration limit; mpc.cpp:61-68 publishes only on 0), 2 from "qp_status"."""
    g, name = golden_rti, "circle_N80"
    env = dict(ENV)
    env.pop("BROV_SHIM_MAXITER_STATUS", None)
    if strict:
        env["BROV_SHIM_MAXITER_STATUS"] = "2"
    out = _run(san_bins[0], _blob(tmp_path, g, name, 2), "M", env=env)
    m = [ln for ln in out.splitlines() if ln.startswith("MAXITER")][0].split()
    assert (int(m[2]), int(m[4]), int(m[6]), int(m[8])) == ((2, 2, 2, 1) if strict else (0, 0, 2, 1))
    u0 = np.array([float(v) for v in m[10:14]])
    assert np.abs(u0).max() <= 8.0 and np.abs(np.abs(u0) - 8.0).max() < 1e-9    # the truncated step: every input at its bound here


def test_oracle_batch_driver_under_sanitizers(san_bins, tmp_path, golden_traj, oracle):
    """orc_rti_step_batch (OpenMP, 4 threads) on 48 instances at N = 20, a third of them metres off (active bounds: tries + interior-point
    iterations), three ticks -- under the sanitizers, and equal to the optimised library's records"""
    N, nb, nt = 20, 48, 3
    circ = golden_traj["circle"]
    rng = np.random.default_rng(12)
    x0 = np.zeros((nb, 12)); x0[:, :6] = circ[0, :6]
    x0 += rng.normal(size=(nb, 12)) * 0.05
    x0[: nb // 3, :3] += rng.uniform(-4, 4, (nb // 3, 3))
    from bluerov2_amd.solver import P_NOMINAL
    p = np.ascontiguousarray(P_NOMINAL, dtype=np.float64)
    blob = np.concatenate([[N, nb, nt, 4], x0.ravel(), p] + [circ[k:k + N + 1].ravel() for k in range(nt)])
    inp = tmp_path / "batch.bin"
    inp.write_bytes(blob.astype(np.float64).tobytes())
    out = _run(san_bins[1], str(inp))
    recs = np.array([[float(v) for v in ln.split()[2:]] for ln in out.splitlines() if ln.startswith("REC")]).reshape(nt, nb, 8)
    op = oracle.opts(N, 1.0 / N)
    x, u, pi, lam = oracle.init_iterate(op, nb)
    pp = np.ascontiguousarray(np.broadcast_to(p, (nb, N + 1, 16)))
    prev = None
    saw_loop = False
    for k in range(nt):
        yref = np.ascontiguousarray(np.broadcast_to(circ[k:k + N + 1], (nb, N + 1, 16)))
        _, prev = oracle.rti_step_batch(op, x0, yref, pp, x, u, pi, lam, res_prev=prev)
        assert np.array_equal(recs[k, :, 6].astype(int), prev["status"]) and np.array_equal(recs[k, :, 7].astype(int), prev["qp_iter"])
        ok = prev["status"] == 0
        assert np.abs(recs[k, ok, :4] - prev["u0"][ok]).max() < 1e-7      # (-O1 without FMA against -O3 -march=x86-64-v3: rounding only)
        saw_loop = saw_loop or (prev["qp_iter"] > 0).any()
    assert saw_loop
