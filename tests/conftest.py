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


@pytest.fixture(scope="session", autouse=True)
def _oracle_threads():
    """The CPU oracle is the checker of most GPU tests, and torch's default intra-op pool on a GPU box is 128 threads: one P-frame at 512x768
    takes 29 s with it against 11.7 s with 8 or 16 (`tools/oracle_threads.py`: small convs drown in the fork / join).  16 = the CPU share
    of a one-GPU box, what bench.py's cpu_baseline uses as well."""
    import torch
    torch.set_num_threads(min(16, torch.get_num_threads()))
    yield
