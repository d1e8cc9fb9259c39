import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pmv():
    """the product package (directory name has a hyphen, so import it by string)"""
    return importlib.import_module("practical-multi-view_amd")


@pytest.fixture(scope="session")
def orc():
    import orc_binding
    return orc_binding.load()


@pytest.fixture(scope="session")
def gpu_ctx_factory(pmv):
    made = []

    def make(w, h, **kw):
        c = pmv.Context(w, h, **kw)
        made.append(c)
        return c

    yield make
    for c in made:
        c.close()
