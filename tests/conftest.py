"""pytest configuration: markers and shared fixtures.

`-m "not gpu"` : oracle vs golden vectors, host logic, ABI surface (no GPU needed).
`-m gpu`       : parity tests proper — HIP path through the C ABI vs the oracle.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """ctypes handle on oracle/liboracle.so (built on demand with make)."""
    from tests.oracle_lib import load_oracle

    return load_oracle()


@pytest.fixture(scope="session")
def gpu_lib():
    """The product library; fails loudly (no CPU fallback) when it cannot load."""
    from codesearch_amd import _lib

    return _lib.load()
