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


def _cpu_share() -> int:
    """Threads this job may really use: the cgroup CPU quota and the affinity mask, not the cores the host shows (a
    one-GPU box shows 128 and grants 16: an OpenMP oracle started with 128 threads runs 3-8x slower)."""
    import math

    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, math.ceil(int(q) / int(p))))
    except (OSError, ValueError, IndexError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, math.ceil(q / p)))
        except (OSError, ValueError):
            pass
    return max(1, n)


os.environ.setdefault("OMP_NUM_THREADS", str(_cpu_share()))  # before liboracle.so (OpenMP) is loaded


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


@pytest.fixture
def lab_lib(monkeypatch):
    """libcsgpu_diag.so standing in for the product library for ONE test: FastEmbedder / VectorStore objects created
    inside the test bind to it.  Laboratory knobs (cs_lab_env in csrc/common.hpp: tile shapes, rejected kernel variants,
    fault injection) and the variants themselves (the one-launch forward, the 32x32x16 wide GEMM, the non-default
    attention loops) exist only there; the product library reads none of them."""
    from codesearch_amd import _lib

    diag = _lib.load_diag()
    monkeypatch.setattr(_lib, "_LIB", diag)
    return diag


@pytest.fixture(scope="session")
def diag_lib():
    """libcsgpu_diag.so: the product sources built with -DCS_DIAGNOSTICS, which adds the operator-level cs_debug_* entry
    points (include/codesearch_gpu_diag.h) the unit parity tests of single kernels go through."""
    from codesearch_amd import _lib

    return _lib.load_diag()
