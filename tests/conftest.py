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
def golden_traj():
    return np.load(os.path.join(GOLDEN, "traj_head.npz"))


def scenario_names(g):
    return sorted(set(k.split("/")[0] for k in g.files))


def scenario_ticks(g, name):
    k = 0
    while f"{name}/x{k}" in g.files:
        k += 1
    return k
