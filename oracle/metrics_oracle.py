"""CPU restatement of the reference's structural-loss CUDA extension.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED for the approximate-EMD functions of this file (the nearest-neighbour / Chamfer functions are pinned to the
reference's own pure-torch distChamfer through tests/golden/g14_evaluation.npz): the reference implements these operators only as CUDA sources
(lib/metrics/pytorch_structural_losses/src/{nndistance.cu, approxmatch.cu}; nvcc is not in this image, so the extension
is unbuildable here) and holds no golden vectors or tests for them.  The functions below restate the published
algorithm of those kernels in float32 numpy, each citing the lines it follows; tests additionally anchor them on
independent facts (brute-force nearest neighbours in float64, the reference's own pure-torch Chamfer formula
evaluation_metrics.py:35-45, exact optimal transport from scipy for small sets, finite differences for the gradients).

Only tests/ may import this module.
"""
import numpy as np

F = np.float32


def _sqdist(a, b):
    """[n,3],[m,3] -> [m,n] float32 squared distances, summed x,y,z in that order (nndistance.cu:27-33, no FMA)."""
    d = b[:, None, :].astype(F) - a[None, :, :].astype(F)
    return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]


def nn_distance(xyz1, xyz2):
    """nndistance.cu:2-124 / structural_loss.cpp:82-104.  (b,n,3),(b,m,3) -> dist1 (b,n), idx1, dist2 (b,m), idx2.
    First minimiser wins (strict '<' in scan order = numpy argmin)."""
    b = xyz1.shape[0]
    d1, i1, d2, i2 = [], [], [], []
    for i in range(b):
        D = _sqdist(xyz1[i], xyz2[i])          # [m][n]
        i1.append(D.argmin(0)); d1.append(D.min(0))
        i2.append(D.argmin(1)); d2.append(D.min(1))
    return (np.stack(d1).astype(F), np.stack(i1).astype(np.int32), np.stack(d2).astype(F), np.stack(i2).astype(np.int32))


def nn_distance_grad(xyz1, xyz2, g1, idx1, g2, idx2):
    """nndistance.cu:129-169: both directions scatter into both gradients."""
    ga = np.zeros(xyz1.shape, np.float64)
    gb = np.zeros(xyz2.shape, np.float64)
    for i in range(xyz1.shape[0]):
        t = 2.0 * g1[i][:, None].astype(np.float64) * (xyz1[i].astype(np.float64) - xyz2[i][idx1[i]])
        ga[i] += t
        np.add.at(gb[i], idx1[i], -t)
        t = 2.0 * g2[i][:, None].astype(np.float64) * (xyz2[i].astype(np.float64) - xyz1[i][idx2[i]])
        gb[i] += t
        np.add.at(ga[i], idx2[i], -t)
    return ga.astype(F), gb.astype(F)


def approx_match(xyz1, xyz2):
    """approxmatch.cu:3-182.  (b,n,3),(b,m,3) -> match (b,m,n): nine auction levels -4^j, j = 7..-1."""
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    multiL, multiR = (F(1), F(n // m)) if n >= m else (F(m // n), F(1))           # :6-12
    out = np.zeros((b, m, n), F)
    for i in range(b):
        D = _sqdist(xyz1[i], xyz2[i])                                                # [m][n]
        remainL = np.full(n, multiL, F)
        remainR = np.full(m, multiR, F)
        match = out[i]
        for j in range(7, -2, -1):
            level = F(-(4.0 ** j))
            E = np.exp(level * D, dtype=F)                                           # [m][n]
            suml = F(1e-9) + (E * remainR[:, None]).sum(0, dtype=F)                  # :33-61
            ratioL = remainL / suml
            sumr = (E * ratioL[None, :]).sum(1, dtype=F) * remainR                   # :63-96
            consumption = np.minimum(remainR / (sumr + F(1e-9)), F(1.0))
            ratioR = consumption * remainR
            remainR = np.maximum(F(0), remainR - sumr)
            W = E * ratioL[None, :] * ratioR[:, None]                                # :98-150
            match += W
            remainL = np.maximum(F(0), remainL - W.sum(0, dtype=F))
    return out


def match_cost(xyz1, xyz2, match):
    """approxmatch.cu:184-224: out[i] = sum match * euclidean distance."""
    return np.stack([(match[i].astype(np.float64) * np.sqrt(_sqdist(xyz1[i], xyz2[i]).astype(np.float64))).sum()
                     for i in range(xyz1.shape[0])]).astype(F)


def match_cost_grad(xyz1, xyz2, match):
    """approxmatch.cu:229-297: gradients w.r.t. both sets for a FIXED matching (distance clamped at 1e-10)."""
    g1 = np.zeros(xyz1.shape, np.float64)
    g2 = np.zeros(xyz2.shape, np.float64)
    for i in range(xyz1.shape[0]):
        diff = xyz1[i][None, :, :].astype(np.float64) - xyz2[i][:, None, :]          # [m][n][3] = a_k - b_l
        dist = np.maximum(np.sqrt((diff ** 2).sum(-1)), 1e-10)
        w = (match[i] / dist)[..., None] * diff
        g1[i] = w.sum(0)
        g2[i] = -w.sum(1)
    return g1.astype(F), g2.astype(F)


def chamfer_bmm(x, y):
    """The reference's own pure-torch Chamfer (evaluation_metrics.py:35-45: |x|^2 + |y|^2 - 2 x.y), float64 here."""
    x = x.astype(np.float64); y = y.astype(np.float64)
    P = (x ** 2).sum(-1)[:, :, None] + (y ** 2).sum(-1)[:, None, :] - 2 * np.einsum('bnd,bmd->bnm', x, y)
    return P.min(2), P.min(1)      # dist1 (b,n): x -> nearest y ; dist2 (b,m)
