import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def orc():
    """The parity oracle (oracle/oracle.py), built on demand with gcc."""
    from oracle import oracle

    oracle.build()
    # OpenMP would start one thread per core it can SEE; under a cgroup quota that is many more than it may use, and the
    # small cases here then spend their time in thread start-up.  The quota if there is one, else the affinity mask.
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    oracle.set_num_threads(n)
    return oracle
