import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver at round end)")


@pytest.fixture(scope="session")
def orc64():
    from oracle import oracle
    oracle.build()
    return oracle.Oracle("f64")


@pytest.fixture(scope="session")
def orc80():
    from oracle import oracle
    oracle.build()
    return oracle.Oracle("f80")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return {n[:-4]: np.load(os.path.join(GOLDEN, n)) for n in os.listdir(GOLDEN) if n.endswith(".npz")}
