import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle (torch intra-op threads) runs on the cores this process may use, not on every logical CPU of the
    # host: a GPU box gives one job a 16-core share of a much larger machine, and an OpenMP team sized for the whole
    # machine spends the oracle's time in contention (the R101 whole-iteration test took 112 ... 150 s by box).
    if not os.environ.get("OMP_NUM_THREADS"):
        try:
            import torch
            n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            torch.set_num_threads(max(1, min(n, 16)))
            os.environ["OMP_NUM_THREADS"] = str(max(1, min(n, 16)))      # (child processes of the multi-rank tests inherit it)
        except Exception:
            pass


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
