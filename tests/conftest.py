import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(autouse=True)
def _bounded_cpu_threads():
    """The oracle side of most tests is thousands of tiny CPU ops: on a many-core box torch's default thread pool
    makes them several times slower (hand-off per op) and their run time erratic.  Eight threads keep the few
    real-size oracle steps (N = 256 ... 512, 6890 vertices) fast enough; the small cases gain."""
    import torch
    n = torch.get_num_threads()
    torch.set_num_threads(min(n, 8))
    yield
    torch.set_num_threads(n)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + '.npz')))


def _np(x):
    if hasattr(x, 'detach'):
        x = x.detach().cpu().numpy()
    elif isinstance(x, (list, tuple)) and len(x) and hasattr(x[0], 'detach'):
        x = [float(t) for t in x]
    return np.asarray(x, dtype=np.float64)


def rel_err(a, b):
    """max |a-b| / max(|b|) -- relative to the tensor's scale (fp32 parity metric)."""
    a = _np(a)
    b = _np(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    denom = max(np.abs(b).max(), 1e-30)
    return float(np.abs(a - b).max() / denom)
