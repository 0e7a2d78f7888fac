"""Structural losses between point sets: Chamfer nearest-neighbour distances and the approximate earth-mover cost.

Host-side mirror of lib/metrics/pytorch_structural_losses/{nn_distance.py:5-37, match_cost.py:5-45} and the helpers
that wrap them (lib/metrics/evaluation_metrics.py:21-30, lib/networks/utils.py:34-42).  Same names, argument meaning
and return values; the compute runs in csrc/gwtf_metrics.hip through the C ABI (include/gwtf.h).  Inputs must be float32,
contiguous and on a HIP device -- the checks the reference's extension applies (structural_loss.cpp:10-12); there is no
CPU path.
"""
import torch
from torch.autograd import Function

from . import _lib
from ._lib import GwtfError, _ptr, _stream, check


def _sets(seta, setb):
    if seta.dim() != 3 or setb.dim() != 3 or seta.shape[2] != 3 or setb.shape[2] != 3 or seta.shape[0] != setb.shape[0]:
        raise GwtfError(f'point sets must be (b,n,3) and (b,m,3): got {tuple(seta.shape)} and {tuple(setb.shape)}')
    _ptr(seta, 'seta'), _ptr(setb, 'setb')        # device / contiguity / dtype checks before anything touches HIP
    if min(seta.shape[0], seta.shape[1], setb.shape[1]) == 0:
        raise GwtfError('empty point set')
    return seta.shape[0], seta.shape[1], setb.shape[1]


def nn_distance_raw(seta, setb):
    """NNDistance (structural_loss.cpp:82-104): dist1 (b,n), idx1 (b,n) int32, dist2 (b,m), idx2 (b,m) int32."""
    b, n, m = _sets(seta, setb)
    L = _lib.lib()
    dev = seta.device
    dist1 = torch.empty(b, n, device=dev, dtype=torch.float32)
    dist2 = torch.empty(b, m, device=dev, dtype=torch.float32)
    idx1 = torch.empty(b, n, device=dev, dtype=torch.int32)
    idx2 = torch.empty(b, m, device=dev, dtype=torch.int32)
    with torch.cuda.device(dev):
        check(L.gwtf_nn_distance(_ptr(seta, 'seta'), _ptr(setb, 'setb'), dist1.data_ptr(), idx1.data_ptr(),
                                 dist2.data_ptr(), idx2.data_ptr(), b, n, m, _stream(seta)))
    return dist1, idx1, dist2, idx2


class NNDistanceFunction(Function):
    """nn_distance.py:5-35: forward returns (dist1, dist2); the argmin indices are kept for the backward."""

    @staticmethod
    def forward(ctx, seta, setb):
        dist1, idx1, dist2, idx2 = nn_distance_raw(seta, setb)
        ctx.save_for_backward(seta, setb)
        ctx.idx1, ctx.idx2 = idx1, idx2
        return dist1, dist2

    @staticmethod
    def backward(ctx, grad_dist1, grad_dist2):
        seta, setb = ctx.saved_tensors
        b, n, m = _sets(seta, setb)
        grada, gradb = torch.empty_like(seta), torch.empty_like(setb)
        with torch.cuda.device(seta.device):
            check(_lib.lib().gwtf_nn_distance_grad(
                _ptr(seta, 'seta'), _ptr(setb, 'setb'), _ptr(grad_dist1.contiguous(), 'grad_dist1'), ctx.idx1.data_ptr(),
                _ptr(grad_dist2.contiguous(), 'grad_dist2'), ctx.idx2.data_ptr(), _ptr(grada, 'grada'),
                _ptr(gradb, 'gradb'), b, n, m, _stream(seta)))
        return grada, gradb


nn_distance = NNDistanceFunction.apply


def approx_match(seta, setb):
    """ApproxMatch (structural_loss.cpp:22-37): match (b,m,n) and the (b,2(n+m)) scratch the reference also returns."""
    b, n, m = _sets(seta, setb)
    match = torch.empty(b, m, n, device=seta.device, dtype=torch.float32)
    temp = torch.empty(b, (n + m) * 2, device=seta.device, dtype=torch.float32)
    with torch.cuda.device(seta.device):
        check(_lib.lib().gwtf_approx_match(_ptr(seta, 'seta'), _ptr(setb, 'setb'), _ptr(match, 'match'),
                                           _ptr(temp, 'temp'), b, n, m, _stream(seta)))
    return match, temp


class MatchCostFunction(Function):
    """match_cost.py:5-43: cost (b,) of the approximate matching; the matching is treated as constant in backward."""

    @staticmethod
    def forward(ctx, seta, setb):
        b, n, m = _sets(seta, setb)
        cost = torch.empty(b, device=seta.device, dtype=torch.float32)
        if not (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            # evaluation (the reference's only use, evaluation_metrics.py:25-30): the fused schedule, no (b,m,n) matrix
            temp = torch.empty(b, (n + m) * 2, device=seta.device, dtype=torch.float32)
            with torch.cuda.device(seta.device):
                check(_lib.lib().gwtf_emd_cost(_ptr(seta, 'seta'), _ptr(setb, 'setb'), _ptr(temp, 'temp'),
                                               _ptr(cost, 'cost'), b, n, m, _stream(seta)))
            return cost
        match, _ = approx_match(seta, setb)
        with torch.cuda.device(seta.device):
            check(_lib.lib().gwtf_match_cost(_ptr(seta, 'seta'), _ptr(setb, 'setb'), _ptr(match, 'match'),
                                             _ptr(cost, 'cost'), b, n, m, _stream(seta)))
        ctx.save_for_backward(seta, setb)
        ctx.match = match
        return cost

    @staticmethod
    def backward(ctx, grad_output):
        seta, setb = ctx.saved_tensors
        b, n, m = _sets(seta, setb)
        grada, gradb = torch.empty_like(seta), torch.empty_like(setb)
        with torch.cuda.device(seta.device):
            check(_lib.lib().gwtf_match_cost_grad(_ptr(seta, 'seta'), _ptr(setb, 'setb'), _ptr(ctx.match, 'match'),
                                                  _ptr(grada, 'grada'), _ptr(gradb, 'gradb'), b, n, m, _stream(seta)))
        g = grad_output.unsqueeze(1).unsqueeze(2)
        return grada * g, gradb * g


match_cost = MatchCostFunction.apply


def distChamferCUDA(x, y):
    """evaluation_metrics.py:21-22 / utils.py:34-35 (name kept so the reference's callers need no edit)."""
    return nn_distance(x, y)


def emd_approx(sample, ref):
    """evaluation_metrics.py:25-30: approximate EMD per cloud, normalised by the number of points."""
    N, N_ref = sample.size(1), ref.size(1)
    assert N == N_ref, "Not sure what would EMD do in this case"
    return match_cost(sample, ref) / float(N)


def f_score(predicted_clouds, true_clouds, threshold=0.001):
    """utils.py:38-42."""
    ld, rd = distChamferCUDA(predicted_clouds, true_clouds)
    precision = 100. * (rd < threshold).float().mean(1)
    recall = 100. * (ld < threshold).float().mean(1)
    return 2. * precision * recall / (precision + recall + 1e-7)
