import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle_ffi
    return oracle_ffi.oracle()


@pytest.fixture(scope="session")
def hp_lib():
    """The product library; built in-tree by __graft_entry__.build()."""
    from hiphase_amd import _ffi
    import __graft_entry__
    __graft_entry__.build()   # no-op when libhiphase_gpu.so is newer than its sources
    return _ffi.lib()


def load_golden(name):
    import json
    with open(os.path.join(ROOT, "tests", "golden", name)) as f:
        return json.load(f)
