"""Per-point linear map and activation used as parameter holders by the flow modules.

Mirrors the public surface of the reference's ``lib/networks/layers.py`` (``SharedDot``
:13-45, ``Swish`` :5-10) so checkpoints and constructor calls keep working.  Inside the
point-flow decoder these modules are never *called*: the fused HIP stack kernel reads
their parameters.  Their ``forward`` exists only for callers outside the hot path
(e.g. the reference's PointNet encoder) and is ordinary torch plumbing.
"""
import math

import torch
import torch.nn as nn


class Swish(nn.Module):
    """x * sigmoid(x)  (reference layers.py:9-10)."""

    def forward(self, x):
        return x * torch.sigmoid(x)


class SharedDot(nn.Module):
    """``y[b,:,n] = W x[b,:,n] (+ bias)`` with one weight shared by all points.

    Parameter shapes follow the reference (layers.py:22-26): ``weight``
    ``(n_channels, out_features, in_features)``, ``bias`` ``(n_channels, out_features)``.
    """

    def __init__(self, in_features, out_features, n_channels, bias=False, init_weight=None, init_bias=None):
        super().__init__()
        self.in_features, self.out_features, self.n_channels = in_features, out_features, n_channels
        self.init_weight, self.init_bias = init_weight, init_bias
        self.weight = nn.Parameter(torch.empty(n_channels, out_features, in_features))
        self.bias = nn.Parameter(torch.empty(n_channels, out_features)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        # reference layers.py:29-38: uniform(+-init_weight) or kaiming-uniform(a=0); bias constant
        with torch.no_grad():
            if self.init_weight:
                self.weight.uniform_(-self.init_weight, self.init_weight)
            else:
                # torch's fan rule on a (C,out,in) tensor gives fan_in = out*in: keep that quirk
                bound = math.sqrt(2.0) * math.sqrt(3.0 / (self.out_features * self.in_features))
                self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                self.bias.fill_(self.init_bias if self.init_bias else 0.0)

    def forward(self, x):
        y = torch.matmul(self.weight, x.unsqueeze(1))
        if self.bias is not None:
            y = y + self.bias.unsqueeze(0).unsqueeze(3)
        return y.squeeze(1)

    def extra_repr(self):
        return f'in={self.in_features}, out={self.out_features}, channels={self.n_channels}, bias={self.bias is not None}'
