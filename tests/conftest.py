import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kats():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def coracle():
    """The C oracle (test infrastructure), built on demand with gcc."""
    from oracle import coracle as C
    C.build()
    C.lib()
    return C


@pytest.fixture(scope="session")
def engine():
    """The product engine on cuda:0.  Fails loudly if libsylow_hip.so or the GPU is missing."""
    import sylow_amd
    return sylow_amd.Engine(0)
