import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """build what is missing (the .so files are not in git): hipcc cross-compiles without a GPU"""
    import subprocess
    lib = os.path.join(ROOT, "mpi-bicgstab_amd", "libbicgstab_hip.so")
    dump = os.path.join(ROOT, "mpi-bicgstab_amd", "host", "bicg_mtx_dump")
    if not (os.path.exists(lib) and os.path.exists(dump)):
        subprocess.call(["make", "-j", "8", "-C", os.path.join(ROOT, "mpi-bicgstab_amd"), "all"])
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.call(["make", "-C", os.path.join(ROOT, "oracle"), "all"])


import pytest


@pytest.fixture(autouse=True)
def _switch_variables_restored():
    """the library's token-list variables (BICG_PLAN / BICG_PERSIST / BICG_TEST, csrc/bicg_knobs.h) are put back after every test:
    tests change single tokens with hipsolver.switches() and need not undo them"""
    names = ("BICG_PLAN", "BICG_PERSIST", "BICG_TEST")
    saved = {k: os.environ.get(k) for k in names}
    yield
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
