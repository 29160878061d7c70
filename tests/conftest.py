import math
import os
import sys


def _cpu_budget():
    """CPUs this process may really use: the affinity mask cut down to the cgroup's CFS quota.  The GPU boxes show 256 hardware threads
    but grant 16 CPUs (cpu.max = 1600000 100000): OpenBLAS then starts 64 threads for the oracle's fp64 products, the quota throttles
    them, and the same suite takes 185 s or 624 s depending on the minute."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith('cpu.max'):
                quota, period = parts[0], float(parts[1])
            else:
                quota = parts[0]
                with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as g:
                    period = float(g.read())
            if quota not in ('max', '-1') and period > 0:
                n = min(n, max(1, int(math.ceil(float(quota) / period))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


CPU_BUDGET = _cpu_budget()
for _v in ('OPENBLAS_NUM_THREADS', 'OMP_NUM_THREADS', 'MKL_NUM_THREADS'):      # before numpy / torch load their thread pools
    os.environ.setdefault(_v, str(CPU_BUDGET))

import numpy as np      # noqa: E402
import pytest           # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (gfx950) device; run with -m gpu')
    # numpy may have been imported by a plugin before this file set the environment: limit the live pools as well
    try:
        from threadpoolctl import threadpool_limits
        config._csi_blas_limit = threadpool_limits(limits=CPU_BUDGET)
    except Exception:       # noqa: BLE001 - a missing threadpoolctl only costs time
        pass


def _gpu_visible():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not skip silently; only when the marker
    # expression does not select gpu tests do they get deselected by pytest itself.
    pass


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session')
def oracle():
    from oracle import csi_oracle
    return csi_oracle


@pytest.fixture(scope='session')
def pkg():
    import dl_channel_estimation_mamimo_amd as m
    return m


def rel_rows(y, ref):
    y = np.asarray(y, dtype=np.float64).reshape(-1, np.asarray(y).shape[-1])
    r = np.asarray(ref, dtype=np.float64).reshape(y.shape)
    return float(np.max(np.linalg.norm(y - r, axis=1) / np.linalg.norm(r, axis=1)))
