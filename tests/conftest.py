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
def oracle():
    """The CPU oracle (test infrastructure): oracle/oracle_py.py over oracle/liboracle.so."""
    from oracle import oracle_py
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def code_a():
    from ldpc_erasure_codes_amd import codes
    return codes.load_builtin(1)


@pytest.fixture(scope="session")
def code_b():
    from ldpc_erasure_codes_amd import codes
    return codes.load_builtin(2)


@pytest.fixture(scope="session")
def code_d():
    from ldpc_erasure_codes_amd import codes
    return codes.load_builtin(0)
