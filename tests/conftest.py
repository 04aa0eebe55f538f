import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """The suites need liboracle.so (CPU) and libstrided_hip.so (loads without a GPU).  Both are
    normally prebuilt by __graft_entry__.build(); build them here if they are missing."""
    import oraclelib
    oraclelib.ensure_built()
    import strided_jl_amd as S
    if not os.path.exists(S._lib.LIB_PATH):
        S.build()
    yield
