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
def ctx(psm):
    c = psm.Context(0)
    yield c
    c.close()
