"""CPU restatement of the reference's global prior flow and Gaussian latent losses.  TEST INFRASTRUCTURE ONLY.

Follows lib/networks/flows.py:163-243, decoders.py:7-38 and losses.py:24-41 in float64 numpy on a plain state_dict.
Parity PINNED by tests/golden/g12_prior.npz (generated from the genuine reference by tests/golden/make_golden.py).
Only tests/ may import this module.
"""
import numpy as np

BN_EPS = 1e-5


def _branch(x, st, prefix, X, training):
    h = x @ st[f'{prefix}T_{X}_0.{X}_mlp0.weight'].astype(np.float64).T
    bn = f'{prefix}T_{X}_0.{X}_mlp0_bn.'
    if training:
        mean, var = h.mean(0), h.var(0)
    else:
        mean, var = st[bn + 'running_mean'].astype(np.float64), st[bn + 'running_var'].astype(np.float64)
    h = (h - mean) / np.sqrt(var + BN_EPS) * st[bn + 'weight'] + st[bn + 'bias']
    h = h / (1.0 + np.exp(-h))
    return h @ st[f'{prefix}T_{X}_0.{X}_mlp1.weight'].astype(np.float64).T + st[f'{prefix}T_{X}_0.{X}_mlp1.bias']


def flow(g, st, prefix, warp, mode, training=False):
    G = g.shape[1]
    keep = [i for i in range(G) if i not in set(warp)]
    eps = float(st[prefix + 'eps'][0])
    logvar, mu = np.zeros_like(g), np.zeros_like(g)
    logvar[:, warp] = np.log(eps + np.exp(_branch(g[:, keep], st, prefix, 'logvar', training)))
    mu[:, warp] = _branch(g[:, keep], st, prefix, 'mu', training)
    out = np.exp(0.5 * logvar) * g + mu if mode == 'direct' else np.exp(-0.5 * logvar) * (g - mu)
    return out, mu, logvar


def warp_sets(G, pattern):
    idx = list(range(G))
    return (idx[::2], idx[1::2]) if pattern == 0 else (idx[:G // 2], idx[G // 2:])


def decoder(g, st, n_flows, mode, training=False):
    """-> gs, mus, logvars (direct-ordered lists of 2*n_flows arrays)."""
    g = g.astype(np.float64)
    G = g.shape[1]
    names = [(f'flows.{i}.nvp{k + 1}.', warp_sets(G, i % 2)[k]) for i in range(n_flows) for k in range(2)]
    out = [None] * len(names)
    cur = g
    order = range(len(names)) if mode == 'direct' else range(len(names) - 1, -1, -1)
    for j in order:
        cur, mu, lv = flow(cur, st, names[j][0], names[j][1], mode, training)
        out[j] = (cur, mu, lv)
    return [o[0] for o in out], [o[1] for o in out], [o[2] for o in out]


def gaussian_flow_nll(samples, mus, logvars):
    z = samples[0]
    return 0.5 * (np.sum(sum(logvars) + (z - mus[0]) ** 2 / np.exp(logvars[0])) / z.shape[0] + np.log(2 * np.pi) * z.shape[1])


def gaussian_entropy(logvars):
    return 0.5 * (logvars.shape[1] * (1.0 + np.log(2 * np.pi)) + logvars.sum(1).mean())
