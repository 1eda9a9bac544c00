import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")


@pytest.fixture(scope="session")
def psm():
    return importlib.import_module("prismarine-core_amd")


@pytest.fixture(scope="session")
def scenes():
    return importlib.import_module("prismarine-core_amd.scenes")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def _session_ctx(psm):
    c = psm.Context(0)
    yield c
    c.close()


@pytest.fixture
def ctx(psm, _session_ctx):
    # one context for the session; every test starts with the default sort (a hybrid sort that overflowed in an earlier
    # test has sent the context back to the eight-pass sort: psm_sort_set_algorithm re-arms it)
    psm.RadixSort(_session_ctx).setAlgorithm(2)
    return _session_ctx
