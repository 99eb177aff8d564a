"""Shared fixtures.  `gpu` marks tests that need a real MI355X; everything else
runs on CPU.  The oracle (oracle/) is the checker; rtamd is the product binding."""
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, os.path.join(REPO, "ray-tracer_amd", "python"))
sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(REPO, "tests", "golden", "golden_v1.npz")
    return np.load(path, allow_pickle=False)


@pytest.fixture(scope="session")
def oracle():
    from oracle_py import Oracle, ORACLE_PATH
    if not os.path.exists(ORACLE_PATH):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle"), "liboracle.so"])
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    """The real reference build, when oracle/_ref/libref.so exists."""
    from oracle_py import Ref
    if not Ref.available():
        pytest.skip("oracle/_ref/libref.so not built (needs /root/reference)")
    return Ref()
