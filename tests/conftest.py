"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path.

`-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI symbol export (no GPU needed).
`-m gpu`     : parity tests proper -- the HIP path through the C-ABI against the oracle / goldens.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def built_artefacts():
    """Built files are not in the history (they travel with the working tree).  On a tree that lacks them -- a fresh
    checkout -- compile what the tests load: the HIP library (hipcc cross-compiles gfx950 without a GPU) and the checker."""
    import subprocess
    pkg = os.path.join(ROOT, "data-compressor_amd")
    if not os.path.exists(os.path.join(pkg, "libdega_hip.so")) and os.path.exists("/opt/rocm/bin/hipcc"):
        subprocess.run(["make", "-s", "-C", os.path.join(pkg, "csrc")], check=True)
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True)
    yield
