"""CPU checks of oracle/metrics_oracle.py (Chamfer / approximate EMD restatement) against independent facts.

The reference has no fixtures for its CUDA structural losses; the Chamfer half is pinned through the reference's pure-torch
distChamfer (golden g14), the approximate-EMD half stays unpinned (oracle header).  Further anchors: float64 brute force, the reference's own pure-torch Chamfer formula (evaluation_metrics.py:35-45), exact optimal
transport for small sets, conservation of mass in the auction, and finite differences.
"""
import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment

from oracle import metrics_oracle as mo


def clouds(seed, b, n, m, scale=0.5):
    r = np.random.default_rng(seed)
    return (r.standard_normal((b, n, 3)) * scale).astype(np.float32), (r.standard_normal((b, m, 3)) * scale).astype(np.float32)


def test_nn_distance_oracle_matches_the_references_torch_chamfer_g14():
    """Outputs of the reference's own distChamfer (golden g14): distChamfer returns (per point of b, per point of a)."""
    import os
    D = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'g14_evaluation.npz'))
    d1, _, d2, _ = mo.nn_distance(D['a'], D['b'])
    np.testing.assert_allclose(d1, D['chamfer_second'], atol=2e-6)
    np.testing.assert_allclose(d2, D['chamfer_first'], atol=2e-6)


@pytest.mark.parametrize('b,n,m', [(2, 64, 64), (3, 100, 37), (1, 5, 300)])
def test_nn_distance_matches_bruteforce_and_reference_formula(b, n, m):
    x, y = clouds(0, b, n, m)
    d1, i1, d2, i2 = mo.nn_distance(x, y)
    D = ((x[:, :, None, :].astype(np.float64) - y[:, None, :, :]) ** 2).sum(-1)
    np.testing.assert_allclose(d1, D.min(2), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(d2, D.min(1), rtol=1e-5, atol=1e-7)
    # the selected index attains the minimum (ties aside)
    np.testing.assert_allclose(np.take_along_axis(D, i1[:, :, None].astype(np.int64), 2)[..., 0], D.min(2), rtol=1e-5, atol=1e-7)
    c1, c2 = mo.chamfer_bmm(x, y)
    np.testing.assert_allclose(d1, c1, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(d2, c2, rtol=1e-4, atol=1e-5)


def test_nn_distance_first_minimum_wins():
    x = np.zeros((1, 2, 3), np.float32)
    y = np.tile(np.array([[1, 0, 0]], np.float32), (1, 7, 1))       # seven equidistant candidates
    _, i1, _, i2 = mo.nn_distance(x, y)
    assert (i1 == 0).all() and (i2 == 0).all()


def test_nn_distance_grad_finite_difference():
    x, y = clouds(1, 2, 20, 30)
    d1, i1, d2, i2 = mo.nn_distance(x, y)
    r = np.random.default_rng(2)
    g1, g2 = r.standard_normal(d1.shape).astype(np.float32), r.standard_normal(d2.shape).astype(np.float32)
    ga, gb = mo.nn_distance_grad(x, y, g1, i1, g2, i2)

    def loss(xx, yy):
        D = ((xx[:, :, None, :].astype(np.float64) - yy[:, None, :, :]) ** 2).sum(-1)
        return (D.min(2) * g1).sum() + (D.min(1) * g2).sum()
    h = 1e-4
    for (arr, grad) in ((x, ga), (y, gb)):
        for idx in [(0, 3, 1), (1, 7, 2), (1, 0, 0)]:
            p = arr.copy(); p[idx] += h
            q = arr.copy(); q[idx] -= h
            fd = (loss(p, y) - loss(q, y)) / (2 * h) if arr is x else (loss(x, p) - loss(x, q)) / (2 * h)
            assert abs(fd - grad[idx]) < 2e-3 * max(1, abs(fd))


@pytest.mark.parametrize('n', [16, 64])
def test_approx_match_conserves_mass_and_bounds_exact_emd(n):
    x, y = clouds(3, 2, n, n, scale=0.3)
    match = mo.approx_match(x, y)
    assert (match >= 0).all()
    # every point ends (almost) fully matched; nothing is over-subscribed
    assert match.sum(1).max() <= 1 + 1e-4 and match.sum(2).max() <= 1 + 1e-4
    assert match.sum(1).min() > 0.95 and match.sum(2).min() > 0.95
    cost = mo.match_cost(x, y, match)
    for i in range(2):
        Dm = np.sqrt(((x[i][:, None, :].astype(np.float64) - y[i][None, :, :]) ** 2).sum(-1))
        r, c = linear_sum_assignment(Dm)
        exact = Dm[r, c].sum()
        # a sub-stochastic coupling carrying >= 95 % of the mass cannot be much cheaper than the optimum, and the
        # auction is known to land within a few tens of percent above it
        assert 0.9 * exact <= cost[i] <= 1.6 * exact


def test_approx_match_unequal_sizes_uses_integer_multiplicity():
    x, y = clouds(4, 1, 24, 8, scale=0.3)          # n = 3 m: every right point can absorb 3 units
    match = mo.approx_match(x, y)
    assert match.shape == (1, 8, 24)
    assert match.sum(1).max() <= 1 + 1e-4 and match.sum(2).max() <= 3 + 1e-3
    assert match.sum() > 0.95 * 24


def test_match_cost_grad_finite_difference():
    x, y = clouds(5, 1, 12, 12, scale=0.3)
    match = mo.approx_match(x, y)
    g1, g2 = mo.match_cost_grad(x, y, match)
    h = 1e-3
    for arr, grad, which in ((x, g1, 0), (y, g2, 1)):
        for idx in [(0, 2, 0), (0, 11, 2)]:
            p = arr.copy(); p[idx] += h
            q = arr.copy(); q[idx] -= h
            cp = mo.match_cost(p, y, match) if which == 0 else mo.match_cost(x, p, match)
            cq = mo.match_cost(q, y, match) if which == 0 else mo.match_cost(x, q, match)
            fd = (float(cp[0]) - float(cq[0])) / (2 * h)
            assert abs(fd - grad[idx]) < 5e-3 * max(1, abs(fd))
