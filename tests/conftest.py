import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))
    return load


def usable_cores():
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota (the GPU box runs the suite in a
    container; os.cpu_count() reports the whole host)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                    n = min(n, max(1, int(q / per + 0.5)))
        except (OSError, ValueError, IndexError):
            pass
    return n


@pytest.fixture(scope='session', autouse=True)
def _oracle_cpu_threads():
    """The CPU oracle is stock torch: by default it starts one OpenMP thread per core of the HOST, and on a box whose
    container may use a fraction of them the oracle's many small ops then spend their time in oversubscribed barriers (the
    30-iteration trajectory test took 190 s).  Pin the intra-op pool to the cores we can use (SRHIP_TEST_THREADS overrides)."""
    import torch
    n = int(os.environ.get('SRHIP_TEST_THREADS', '0')) or min(usable_cores(), 32)
    torch.set_num_threads(max(1, n))
    yield
