import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def oracle_cpu_threads():
    """The CPU oracle (torch CPU, fp32 and float64) is what most of the GPU suite's wall time goes into, and torch's default - one thread
    per logical CPU - is its WORST setting on the GPU box's 128-thread hosts (bench.py's sweep: 24 frames/s at 128 threads against ~100 at
    16 - 32).  Sixteen threads for the whole session; results do not depend on it beyond fp32 summation order inside oneDNN, which the
    tolerances already cover (profiles/r05/a_reference_self_consistency.txt measures exactly that spread)."""
    import torch
    before = torch.get_num_threads()
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    yield
    torch.set_num_threads(before)
