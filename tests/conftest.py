import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')

# stated fp32 tolerances (BASELINE.md section 4 / SURVEY 8c): grounded in the reference's own fp32-vs-fp64 noise
TOL_COORD = 2e-5      # absolute, coordinates
TOL_LOGDET = 1e-5     # absolute, per-point per-coordinate sum of logvars
TOL_NLL_REL = 1e-5    # relative, per-shape NLL


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


@pytest.fixture(scope='session')
def have_gpu():
    import torch
    return torch.cuda.is_available()
