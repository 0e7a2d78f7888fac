"""Global prior flow on the shape latent and the Gaussian losses that consume it.

Host-side mirror of lib/networks/flows.py:163-243 (RealNVPFlow, RealNVPFlowCouple), lib/networks/decoders.py:7-38
(GlobalRNVPDecoder) and lib/networks/losses.py:24-41 (GaussianFlowNLL, GaussianEntropy): same constructors, attribute
names, ``state_dict`` keys and list-returning forward, so the reference's checkpoints and callers (models.py:137-151)
work unchanged.  The work is per SHAPE -- B rows of G latents, 14 elementary flows of two (B x G/2)(G/2 x F) GEMM pairs
("tiny") -- so it is a chain of plain library GEMMs on the HIP device (torch -> rocBLAS), not a
hand-written kernel; the two shipped warp patterns (even/odd, halves) are applied as strided slices instead of the
reference's index gathers.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from .layers import Swish


def _as_slice(inds, n):
    """A python slice selecting exactly ``inds`` out of range(n), or None."""
    inds = list(inds)
    if not inds:
        return None
    step = inds[1] - inds[0] if len(inds) > 1 else 1
    if step <= 0:
        return None
    sl = slice(inds[0], inds[-1] + 1, step)
    return sl if list(range(n))[sl] == inds else None


class RealNVPFlow(nn.Module):
    """One affine coupling on the latent (reference flows.py:163-216)."""

    def __init__(self, n_features, g_n_features, weight_std=0.01, warp_inds=[0], eps=1e-6):
        super().__init__()
        self.n_features, self.g_n_features, self.weight_std = n_features, g_n_features, weight_std
        self.warp_inds = [int(i) for i in warp_inds]
        self.keep_inds = [i for i in range(g_n_features) if i not in set(self.warp_inds)]
        self.register_buffer('eps', torch.from_numpy(np.array([eps], dtype=np.float32)))
        for X in ('mu', 'logvar'):
            branch = nn.Sequential(OrderedDict([
                (f'{X}_mlp0', nn.Linear(len(self.keep_inds), n_features, bias=False)),
                (f'{X}_mlp0_bn', nn.BatchNorm1d(n_features)),
                (f'{X}_mlp0_swish', Swish()),
                (f'{X}_mlp1', nn.Linear(n_features, len(self.warp_inds), bias=True))]))
            with torch.no_grad():
                branch[-1].weight.normal_(std=weight_std)
                branch[-1].bias.zero_()
            setattr(self, f'T_{X}_0', branch)
        self._warp_sl = _as_slice(self.warp_inds, g_n_features)
        self._keep_sl = _as_slice(self.keep_inds, g_n_features)

    def _take(self, g, inds, sl):
        return g[:, sl] if sl is not None else g[:, inds]

    def _scatter(self, like, values):
        sl = self._warp_sl
        if sl is not None and sl.step in (None, 1):
            # a contiguous block of warped indices: zero padding = one kernel forward, a narrow backward (an in-place slice
            # assignment costs zeros + copy forward and a clone + copy in its CopySlices backward)
            return torch.nn.functional.pad(values, (sl.start, like.shape[1] - sl.stop))
        out = torch.zeros_like(like)
        if self._warp_sl is not None:
            out[:, self._warp_sl] = values
        else:
            out[:, self.warp_inds] = values
        return out

    def forward(self, g, mode='direct'):
        kept = self._take(g, self.keep_inds, self._keep_sl).contiguous()
        logvar = self._scatter(g, torch.log(self.eps + torch.exp(self.T_logvar_0(kept))))
        mu = self._scatter(g, self.T_mu_0(kept))
        if mode == 'direct':
            g_out = torch.exp(0.5 * logvar) * g + mu
        elif mode == 'inverse':
            g_out = torch.exp(-0.5 * logvar) * (g - mu)
        else:
            raise ValueError(f"mode must be 'direct' or 'inverse', got {mode!r}")
        return g_out, mu, logvar


class RealNVPFlowCouple(nn.Module):
    """Two complementary couplings (reference flows.py:219-243): pattern 0 = even / odd, pattern 1 = first / second half."""

    def __init__(self, n_features, g_n_features, weight_std=0.01, pattern=0):
        super().__init__()
        self.n_features, self.g_n_features, self.weight_std, self.pattern = n_features, g_n_features, weight_std, pattern
        idx = list(range(g_n_features))
        if pattern == 0:
            w1, w2 = idx[::2], idx[1::2]
        elif pattern == 1:
            w1, w2 = idx[:g_n_features // 2], idx[g_n_features // 2:]
        else:
            raise ValueError('pattern must be 0 or 1')
        self.nvp1 = RealNVPFlow(n_features, g_n_features, weight_std=weight_std, warp_inds=w1)
        self.nvp2 = RealNVPFlow(n_features, g_n_features, weight_std=weight_std, warp_inds=w2)

    def forward(self, g, mode='direct'):
        if mode == 'direct':
            g1, mu1, lv1 = self.nvp1(g, mode=mode)
            g2, mu2, lv2 = self.nvp2(g1, mode=mode)
        elif mode == 'inverse':
            g2, mu2, lv2 = self.nvp2(g, mode=mode)
            g1, mu1, lv1 = self.nvp1(g2, mode=mode)
        else:
            raise ValueError(f"mode must be 'direct' or 'inverse', got {mode!r}")
        return [g1, g2], [mu1, mu2], [lv1, lv2]


class GlobalRNVPDecoder(nn.Module):
    """Stack of couples; lists are direct-ordered in both modes (reference decoders.py:7-38)."""

    def __init__(self, n_flows, n_features, g_n_features, weight_std=0.01):
        super().__init__()
        self.n_flows, self.n_features, self.g_n_features, self.weight_std = n_flows, n_features, g_n_features, weight_std
        self.flows = nn.ModuleList([RealNVPFlowCouple(n_features, g_n_features, weight_std=weight_std, pattern=i % 2)
                                    for i in range(n_flows)])

    def forward(self, g, mode='direct'):
        gs, mus, logvars = [], [], []
        cur = g
        order = self.flows if mode == 'direct' else reversed(self.flows)
        for flow in order:
            a, b, c = flow(cur, mode=mode)
            if mode == 'direct':
                gs, mus, logvars = gs + a, mus + b, logvars + c
                cur = gs[-1]
            else:
                gs, mus, logvars = a + gs, b + mus, c + logvars
                cur = gs[0]
        return gs, mus, logvars


class GaussianFlowNLL(nn.Module):
    """reference losses.py:24-33: 0.5 * (sum(sum_j logvars_j + (z - mu0)^2 / exp(logvar0)) / B + G log 2 pi)."""

    def forward(self, samples, mus, logvars):
        z, B, G = samples[0], samples[0].shape[0], samples[0].shape[1]
        total = torch.sum(sum(logvars) + (z - mus[0]) ** 2 / torch.exp(logvars[0])) / B
        return 0.5 * (total + float(np.log(2.0 * np.pi)) * G)


class GaussianEntropy(nn.Module):
    """reference losses.py:36-41."""

    def forward(self, logvars):
        return 0.5 * (logvars.shape[1] * (1.0 + float(np.log(2.0 * np.pi))) + logvars.sum(1).mean())
