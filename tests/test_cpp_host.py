"""The header-only C++ host mirror compiles against the C ABI (CPU check) and reproduces the
reference's VectorStore unit test on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "host_mirror_test.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "host_mirror_test")


def build():
    lib_dir = os.path.join(ROOT, "codesearch_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", SRC, "-o", EXE, f"-L{lib_dir}", "-lcsgpu",
                    f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)


def test_cpp_host_mirror_compiles_and_links(gpu_lib):
    build()
    assert os.path.exists(EXE)


def test_cpp_host_callers_reference_known_answers(gpu_lib):
    """codesearch_callers.hpp (prepare_text, clean_docstring, BatchEmbedder, variant merge, RRF fusion)
    against the reference's own unit-test cases; no GPU involved."""
    build()
    r = subprocess.run([EXE, "cpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host callers ok" in r.stdout


@pytest.mark.gpu
def test_cpp_host_mirror_runs_reference_test():
    build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host mirror ok" in r.stdout
