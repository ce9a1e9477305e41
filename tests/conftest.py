import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG_NAME = "gpu-homomorphic-encryption_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (hyphenated directory name, so imported by string).  The C-ABI library is built on demand
    (a no-op when lib/libfhe_hip.so is newer than its sources, e.g. when it travelled with the snapshot)."""
    mod = importlib.import_module(PKG_NAME)
    mod.build_library()
    return mod


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
