import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (gfx950) device; run with -m gpu')


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
