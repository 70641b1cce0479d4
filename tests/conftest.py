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
    config.addinivalue_line("markers", "timeout: per-test timeout (pytest-timeout)")


def pytest_collection_modifyitems(config, items):
    # a hung kernel or collective must not eat the GPU budget: hard per-test limit (thread method kills the process)
    for it in items:
        if it.get_closest_marker("timeout") is None:
            it.add_marker(pytest.mark.timeout(300, method="thread"))


def load_fixture(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def cell_lines_small():
    return load_fixture("cell_lines_small")


@pytest.fixture(scope="session")
def cell_lines():
    return load_fixture("cell_lines")


@pytest.fixture(scope="session")
def pbmc():
    return load_fixture("pbmc_stim_pcs")
