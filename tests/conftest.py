import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    # the CPU oracle runs thousands of small torch ops: on a many-core host (the GPU boxes have 256 logical CPUs) the default
    # "one thread per core" makes each op pay a fork/join over all of them (measured ~800x slower, profiles/r2_cpu_threads.md)
    import torch
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
