import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def _usable_cpus():
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:                                     # cgroup v2 quota (the GPU box exposes 256 CPUs but grants 16)
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def pytest_sessionstart(session):
    """The C-ABI library is part of the product; (re)build it when sources changed (hipcc cross-compiles on CPU)."""
    from flowmse_amd import build as fb
    fb.build(verbose=False)
    import torch
    torch.set_num_threads(min(32, _usable_cpus()))     # CPU references: oversubscribed threads are 50x slower
