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


def tol_at_depth(C, xmax):
    """Stated tolerances carried to deeper stacks and larger coordinates.  The 2e-5 / 1e-5 figures
    were sized on the 12-coupling M1 stack with |x| <= 6, 4-6x above the reference's own fp32-vs-fp64 noise there.  Rounding
    error of a coordinate grows with the number of couplings it passes and with its magnitude (an fp32 ulp at |x| = 64 is
    7.6e-6), the sum of logvars with the number of terms: coordinates 2e-5 * max(1, C/12) * max(1, |x|max/6), log-det
    1e-5 * max(1, C/12) * max(1, |x|max/24) -- every logvar depends with O(1) sensitivity on the coordinates that enter its
    coupling, so beyond |x| of a few tens the coordinates' rounding error dominates the logvar's own (fp32 evaluation of the
    33-coupling f=37 stack against fp64: 5.5e-5 at |x|max = 111, 6e-6 below |x|max = 60).  Measured on the genuine reference
    at full config depth (golden g15_*, fp32 run against its own fp64 run): up to 1.03e-4 / 1.06e-5 at C = 33, |x|max = 64
    -- the same 4-6x margin below this bar."""
    xmax = float(xmax)
    return (TOL_COORD * max(1.0, C / 12.0) * max(1.0, xmax / 6.0),
            TOL_LOGDET * max(1.0, C / 12.0) * max(1.0, xmax / 24.0))


def record_parity(name, **errors):
    """Measured parity errors, printed (pytest -rP / -s) and appended to gpurun_out/parity_errors.jsonl when writable."""
    import json
    line = json.dumps({'case': name, **{k: float(v) for k, v in errors.items()}})
    print('PARITY', line)
    try:
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, 'gpurun_out', 'parity_errors.jsonl'), 'a') as fh:
            fh.write(line + '\n')
    except OSError:
        pass


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


@pytest.fixture(scope='session')
def have_gpu():
    import torch
    return torch.cuda.is_available()
