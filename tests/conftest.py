import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def report():
    """append-only diagnostics file under gpurun_out/ (merged back from the GPU box)"""
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    f = open(os.path.join(d, "parity_report.txt"), "a")
    yield lambda *a: (print(*a, file=f, flush=True), print(*a))
    f.close()
