import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracle (the checker of the parity tests) is PyTorch on the host cores.  torch's default on the GPU box is 128 threads for
    # 256 cores, where the oracle's small matrices run 9 - 12x SLOWER than on 8 - 16 threads (tools/time_oracle_threads.py:
    # DDIM-10, B=3, T=1800: 15.3 s at the default, 1.2 s at 16) - most of the gpu suite's wall time was that.
    import torch
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
