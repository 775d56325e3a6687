import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_built():
    """Build the CPU oracle (plain C, gcc) once per session."""
    odir = os.path.join(ROOT, "oracle")
    subprocess.run(["make", "-s", "oracle"], cwd=odir, check=True)
    return odir
