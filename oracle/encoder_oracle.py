"""CPU restatement of the reference's PointNet cloud encoder and per-shape heads.  TEST INFRASTRUCTURE ONLY.

Follows lib/networks/encoders.py:9-28 (SharedDot -> BatchNorm1d -> ReLU per layer), models.py:127-128 (max over points)
and encoders.py:31-91 (Linear -> BatchNorm1d -> Swish trunk, Linear heads), in float64 numpy on a plain state_dict.
Parity PINNED: tests/test_oracle_golden.py checks it against tests/golden/g11_encoder.npz, produced by the genuine
reference modules (tests/golden/make_golden.py).  Only tests/ may import this module.
"""
import numpy as np

BN_EPS = 1e-5


def _bn(x, st, prefix, training, axes):
    """nn.BatchNorm1d: eval -> running statistics; train -> biased batch statistics (torch semantics)."""
    shape = [1] * x.ndim
    shape[1] = -1
    if training:
        mean, var = x.mean(axes), x.var(axes)
    else:
        mean, var = st[prefix + 'running_mean'].astype(np.float64), st[prefix + 'running_var'].astype(np.float64)
    y = (x - mean.reshape(shape)) / np.sqrt(var.reshape(shape) + BN_EPS)
    return y * st[prefix + 'weight'].astype(np.float64).reshape(shape) + st[prefix + 'bias'].astype(np.float64).reshape(shape)


def pointnet_features(x, st, n_layers, training=False, prefix='features.'):
    """(B,3,N) -> (B,C_last,N).  n_layers = number of SharedDot layers after init_sd."""
    h = x.astype(np.float64)
    for name in ['init_sd'] + [f'sd{i}' for i in range(n_layers)]:
        W = st[prefix + name + '.weight'].astype(np.float64)[0]        # (out, in)
        h = np.einsum('oi,bin->bon', W, h)
        h = np.maximum(_bn(h, st, prefix + name + '_bn.', training, (0, 2)), 0.0)
    return h


def pointnet_pooled(x, st, n_layers, training=False, prefix='features.'):
    return pointnet_features(x, st, n_layers, training, prefix).max(2)


def feature_encoder(x, st, n_layers, deterministic, training=False, prefix=''):
    h = x.astype(np.float64)
    for i in range(n_layers):
        h = h @ st[f'{prefix}features.mlp{i}.weight'].astype(np.float64).T
        if f'{prefix}features.mlp{i}_bn.weight' in st:
            h = _bn(h, st, f'{prefix}features.mlp{i}_bn.', training, (0,))
        h = h / (1.0 + np.exp(-h))
    mu = h @ st[prefix + 'mus.mu_mlp0.weight'].astype(np.float64).T + st[prefix + 'mus.mu_mlp0.bias']
    if deterministic:
        return mu
    return mu, h @ st[prefix + 'logvars.logvar_mlp0.weight'].astype(np.float64).T + st[prefix + 'logvars.logvar_mlp0.bias']
