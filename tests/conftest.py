import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The fp64 oracle is eager PyTorch on tiny tensors (a Python loop over solver steps with autograd): with one thread per core of a
    # many-core GPU host every op pays the thread pool's fan-out and the suite crawls (dopri5 cases: 60 s each on the GPU box against
    # 4 s on 8 cores).  Eight threads is the fastest setting bench.py's cpu_baseline leg finds on those hosts as well.
    import torch
    torch.set_num_threads(min(8, os.cpu_count() or 8))


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
