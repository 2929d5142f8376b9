import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# ---- record of what the parity rules excused --------------------------------------------------------------------------------
# `pytest -q` swallows the prints of status_agreement / values_agree / u0_abs_ok, so every call also appends one line here and the
# session writes the lot to gpurun_out/parity_excused.json (scripts/collect_profiles.py copies its summary into profiles/): a
# count creeping from 0 to the allowance shows in a committed file, not only in a log nobody reads.
_PARITY_LOG = []
SUITE_MAX_STATUS_AMBIGUOUS = 30     # measured 23 (rounds 5 and 6)
SUITE_MAX_VALUES_DIVERGED = 4       # measured 0
SUITE_MAX_U0_EXCUSED = 2            # measured 0


def _parity_note(rule, what, n_checked, n_excused, **detail):
    _PARITY_LOG.append(dict(rule=rule, what=repr(what), checked=int(n_checked), excused=int(n_excused),
                            **{k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in detail.items()}))


def pytest_sessionfinish(session, exitstatus):
    if not _PARITY_LOG:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    by_rule = {}
    for e in _PARITY_LOG:
        r = by_rule.setdefault(e["rule"], dict(calls=0, checked=0, excused=0, max_excused_per_call=0, calls_with_excused=0))
        r["calls"] += 1; r["checked"] += e["checked"]; r["excused"] += e["excused"]
        r["max_excused_per_call"] = max(r["max_excused_per_call"], e["excused"]); r["calls_with_excused"] += int(e["excused"] > 0)
    u0 = [e for e in _PARITY_LOG if e["rule"] == "u0_abs"]
    summary = dict(exitstatus=int(exitstatus), by_rule=by_rule,
                   u0_abs_worst_held_error=max([e["worst_held"] for e in u0], default=None),
                   u0_abs_kkt_histogram_of_held_instances=(np.sum([e["kkt_hist"] for e in u0], axis=0).tolist() if u0 else None),
                   kkt_histogram_edges=KKT_EDGES,
                   bvls_abs=[dict(what=e["what"], instances_x_ticks=e["checked"], kkt_hist=e["kkt_hist"], worst_u=e["worst_u"],
                                  worst_u0=e["worst_u0"], active_bounds=e["active_bounds"]) for e in _PARITY_LOG if e["rule"] == "bvls_abs_1e-8"],
                   entries_with_excused=[e for e in _PARITY_LOG if e["excused"] > 0][:200])
    # Suite-wide ceilings (round 6): the per-call allowances above are what ONE comparison may excuse, these are what
    # the WHOLE session may -- pinned to what the suite shows, so that a regression of one digit turns the run red although every single
    # call stays inside its allowance.  Measured over the sessions of rounds 5-6: 23 status-ambiguous instance-ticks of 497 k (all GPU = 4
    # against oracle = 0 / 2 at entering KKT > 1e6), 0 values outside the scaled tolerance, 0 excused u0.
    ceilings = {"status": SUITE_MAX_STATUS_AMBIGUOUS, "values_scaled": SUITE_MAX_VALUES_DIVERGED, "u0_abs": SUITE_MAX_U0_EXCUSED}
    over = {r: (by_rule[r]["excused"], c) for r, c in ceilings.items() if r in by_rule and by_rule[r]["excused"] > c}
    summary["suite_ceilings"] = dict(ceilings=ceilings, exceeded=over)
    with open(os.path.join(out, "parity_excused.json"), "w") as f:
        json.dump(summary, f, indent=1)
    if over:
        sys.stderr.write(f"\nPARITY ALLOWANCES EXCEEDED SUITE-WIDE (excused, ceiling): {over} -- see gpurun_out/parity_excused.json\n")
        session.exitstatus = pytest.ExitCode.TESTS_FAILED


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle_ffi import Oracle, build
    build()
    return Oracle()


@pytest.fixture(scope="session")
def golden_model():
    return np.load(os.path.join(GOLDEN, "model_vectors.npz"))


@pytest.fixture(scope="session")
def golden_rti():
    return np.load(os.path.join(GOLDEN, "rti_known_answers.npz"))


@pytest.fixture(scope="session")
def golden_rti_options():
    """known answers with non-default weights / boxes / per-stage parameters (scripts/make_golden.py options)"""
    return np.load(os.path.join(GOLDEN, "rti_known_answers_options.npz"))


def scenario_options(g, name):
    """the solver options a fixture scenario carries (empty for the shipped-options fixture)"""
    return {k: list(g[f"{name}/{k}"]) for k in ("W", "We", "lbu", "ubu") if f"{name}/{k}" in g.files}


@pytest.fixture(scope="session")
def golden_traj():
    return np.load(os.path.join(GOLDEN, "traj_head.npz"))


def scenario_names(g):
    return sorted(set(k.split("/")[0] for k in g.files))


def scenario_ticks(g, name):
    k = 0
    while f"{name}/x{k}" in g.files:
        k += 1
    return k


def status_agreement(r_status, o_status, o_kkt, max_ambiguous=4):
    """Status parity rule shared by the batch tests.  Wherever the step is numerically meaningful (entering KKT <= 1e6) the GPU
    and the oracle must report the same status -- no exceptions.  Once an iterate has diverged (KKT > 1e6: QP data of size
    1e6..1e17; full-step SQP has no globalisation, as in the reference) the Riccati recursion works on numbers whose rounding
    errors exceed the input weights, and whether / where the step is declared failed -- a pivot block that stops being positive
    definite (4), the iteration limit (2), a NaN (1), or not at all -- depends on the summation order (MFMA tiles vs scalar
    loops).  At most `max_ambiguous` such instances may disagree, and their meaningless iterates are not compared on that tick.
    Round 5: the disagreements have ONE direction and one cause, and the rule holds them to it.  The kernels take the Cholesky pivot
    form (which the oracle always uses) only while the entering KKT is <= 1e6; above it they keep the explicit 2 x 2-block inverse,
    which loses positive definiteness earlier -- so the GPU reports QP failure (4) on steps the oracle still factorises (0, rarely 2).
    Measured (profiles/r5_status_direction.txt): with the limit lifted (BROV_ROBUST_PIVOT=3) 20 of the suite's 24 disagreements vanish,
    and the mixed batch / the config-4 shard lose 29 % / 23 % to diverged instances grinding through the iteration limit.  The
    allowance is what the suite shows, not a round number: 3 per call in round 5; 4 since round 6, whose factor stage forms the pivot
    inverse in another order of operations (qp/sweeps.hpp, kR6) -- one more instance of ONE call (the config-4 shard's 26-tick run, instances
    123 / 1057 / 2079 / 3405 of 4096 at entering KKT > 1e6) loses positive definiteness a tick earlier, same direction, while the suite-wide
    count stayed where it was (23 -> 23 +- 1 of 497 k; the session hook holds it to <= 30).
    Returns the mask of instances whose values are to be compared (all but the ambiguous ones)."""
    r_status, o_status, o_kkt = np.asarray(r_status), np.asarray(o_status), np.asarray(o_kkt)
    mism = r_status != o_status
    assert np.all((o_kkt[mism] > 1e6) | ~np.isfinite(o_kkt[mism])), (np.nonzero(mism)[0], r_status[mism], o_status[mism], o_kkt[mism])
    assert mism.sum() <= max_ambiguous, (np.nonzero(mism)[0], r_status[mism], o_status[mism])
    if os.environ.get("BROV_ROBUST_PIVOT", "1") == "1" and "BROV_ROBUST_KKT_MAX" not in os.environ:   # (the product's setting)
        assert np.all(r_status[mism] == 4) and np.all((o_status[mism] == 0) | (o_status[mism] == 2)), (
            "a status disagreement in the other direction", np.nonzero(mism)[0], r_status[mism], o_status[mism], o_kkt[mism])
    if mism.sum():
        print(f"[status_agreement] {int(mism.sum())} status-ambiguous instance(s) (entering KKT > 1e6): gpu {r_status[mism]} oracle {o_status[mism]}")
    _parity_note("status", "status", len(mism), mism.sum(), kkt=o_kkt[mism], gpu=r_status[mism], oracle=o_status[mism])
    return ~mism


def values_agree(ok, kkt, what, max_diverged=4, err=None):
    """Value parity rule shared by the batch tests: `ok` is the per-instance verdict of a KKT-scaled comparison
    (|gpu - oracle| <= 1e-7 max(1, KKT)).  Every instance whose step is numerically meaningful (entering KKT <= 1e6) must pass --
    round 3: without the allowance for degenerate bounds round 2 needed (two instances per tick at 1e-4): both sides now end
    their QPs with an exact active-set solve.  A diverged instance (KKT > 1e6, see status_agreement) may miss even the scaled
    tolerance -- its QP is conditioned beyond what FP64 resolves, and both sides may well report success with different garbage --
    but there may be at most `max_diverged` of them per tick; their number is printed so that a creeping regression shows."""
    ok, kkt = np.asarray(ok), np.asarray(kkt)
    bad = ~ok
    assert not np.any(bad & ~(kkt > 1e6)), (what, np.nonzero(bad & ~(kkt > 1e6))[0][:8], kkt[bad][:8],
                                            None if err is None else np.asarray(err)[bad][:8])
    assert bad.sum() <= max_diverged, (what, np.nonzero(bad)[0][:8], kkt[bad][:8])
    if bad.sum():
        print(f"[values_agree] {what}: {int(bad.sum())} diverged instance(s) (KKT > 1e6) outside the scaled tolerance")
    _parity_note("values_scaled", what, len(ok), bad.sum(), kkt=kkt[bad])


KKT_EDGES = [0.0, 1.0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6]
U0_TOL = 1e-5          # BASELINE.json north_star: "matching acados u* within 1e-5"


def u0_abs_ok(u0_gpu, u0_orc, st_gpu, st_orc, kkt, what, max_exceptions=2):
    """The north star's own bar on the one output the node applies: |u0_gpu - u0_oracle|_inf <= 1e-5 ABSOLUTE, whatever the size of
    the instance's QP data.  The KKT-scaled rule above (1e-7 max(1, KKT)) is tighter than this for well-posed instances and looser
    for the saturated / far-off ones that run the QP loop (entering KKT 1e2 .. 1e5 -> 1e-5 .. 1e-2); this rule closes that gap.
    Held: every instance both sides solved (status 0) whose entering KKT is finite and <= 1e6.  No exception at KKT <= 1e5; at most
    `max_exceptions` per call between 1e5 and 1e6 (iterates on their way to divergence: the QP's own conditioning there costs more
    than 11 digits), each printed and recorded.  Returns the per-instance errors."""
    u0_gpu, u0_orc, kkt = np.asarray(u0_gpu), np.asarray(u0_orc), np.asarray(kkt)
    st_gpu, st_orc = np.asarray(st_gpu), np.asarray(st_orc)
    with np.errstate(invalid="ignore"):
        err = np.abs(u0_gpu - u0_orc).reshape(len(kkt), -1).max(axis=1)
        held = (st_gpu == 0) & (st_orc == 0) & np.isfinite(kkt) & (kkt <= 1e6)
        viol = held & ~(err <= U0_TOL)
    hard = viol & (kkt <= 1e5)
    assert not hard.any(), (what, "u0 off by more than 1e-5 absolute", np.nonzero(hard)[0][:8], err[hard][:8], kkt[hard][:8])
    assert viol.sum() <= max_exceptions, (what, np.nonzero(viol)[0][:8], err[viol][:8], kkt[viol][:8])
    for i in np.nonzero(viol)[0]:
        print(f"[u0_abs_ok] {what}: instance {i} |du0| = {err[i]:.2e} at entering KKT {kkt[i]:.2e} (excused: KKT > 1e5)")
    hist = np.histogram(kkt[held], bins=KKT_EDGES)[0]
    _parity_note("u0_abs", what, held.sum(), viol.sum(), worst_held=float(err[held].max()) if held.any() else 0.0,
                 kkt_hist=hist, kkt=kkt[viol], err=err[viol])
    return err
