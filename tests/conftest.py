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


_ORDER = ("test_gpu_parity", "test_gpu_stages", "test_gpu_robust")


def pytest_collection_modifyitems(config, items):
    """The driver runs `pytest -m gpu -x`: parity tests first, then stage and robustness tests, the subprocess fuzz runs last, so
    that one slow or failing fuzz case cannot hide the parity results behind it (stable sort: order inside a file is kept)."""
    def rank(item):
        name = os.path.basename(str(item.fspath))
        for i, stem in enumerate(_ORDER):
            if name.startswith(stem):
                return i
        return len(_ORDER) + (1 if "fuzz" in name else 0)
    items.sort(key=rank)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
