import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


def _usable_cpu_threads():
    """CPU threads this process may really use (cgroup quota, else affinity), capped at the 16 a one-GPU box of the pool gets."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return int(os.environ.get("TAV_CPU_THREADS", min(n, 16)))


@pytest.fixture(scope="session", autouse=True)
def _oracle_cpu_threads():
    """The CPU oracle is most of the GPU suite's wall time (full-depth fp32 forward + backward, up to batch 8).  torch sizes its intra-op pool
    from the HOST's core count, which on a shared GPU box is many times the cgroup quota: cap it at what the process can really run."""
    import torch
    torch.set_num_threads(_usable_cpu_threads())
    yield
