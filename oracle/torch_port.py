"""PyTorch-CPU port of the point-flow decoder forward.  TEST / BASELINE INFRASTRUCTURE ONLY.

BASELINE.json asks for "the reference's PyTorch-CPU path timed on the same box's host cores" as the
reported baseline.  The reference cannot travel to the GPU box, so this file restates its forward with
the same torch CPU ops the reference issues per coupling (matmul, batch_norm, relu, exp, softsign,
sqrt: lib/networks/flows.py:95-117, layers.py:40-45), driven by a plain ``state_dict``.  It is pinned to
the genuine reference through the golden fixtures (tests/test_oracle_golden.py::test_torch_port_*) and
is used ONLY by bench.py's ``cpu_baseline`` leg and by tests -- never by the product path.
"""
import torch
import torch.nn.functional as F

PATTERNS = ((0,), (1,), (2,), (0, 1), (0, 2), (1, 2))


_TRAIN = [False]   # set by decoder_fused(training=...): batch statistics (and in-place running-stat updates on `st`)


def _bn(x, st, prefix, affine):
    return F.batch_norm(x, st[prefix + 'running_mean'], st[prefix + 'running_var'],
                        st.get(prefix + 'weight') if affine else None, st.get(prefix + 'bias') if affine else None,
                        _TRAIN[0], 0.1, 1e-5)


def _film(g, st, prefix, X, which):
    base = f'{prefix}T_{X}_0_cond_{which}.{X}_sd1_film_{which}'
    h = F.linear(g, st[base + '0.weight'])
    h = _bn(h, st, base + '0_bn.', True)
    h = h * torch.sigmoid(h)
    return F.linear(h, st[base + '1.weight'], st[base + '1.bias'])


def _branch(pk, g, st, prefix, X, eps):
    t0 = f'{prefix}T_{X}_0.{X}_'
    h = torch.matmul(st[t0 + 'sd0.weight'], pk.unsqueeze(1)).squeeze(1)
    h = F.relu_(_bn(h, st, t0 + 'sd0_bn.', True))
    h = torch.matmul(st[t0 + 'sd1.weight'], h.unsqueeze(1)).squeeze(1)
    h = _bn(h, st, t0 + 'sd1_bn.', False)
    h = torch.add(eps, torch.exp(_film(g, st, prefix, X, 'w').unsqueeze(2))) * h + _film(g, st, prefix, X, 'b').unsqueeze(2)
    t1 = f'{prefix}T_{X}_1.{X}_sd2.'
    out = torch.matmul(st[t1 + 'weight'], F.relu_(h).unsqueeze(1))
    return out.add_(st[t1 + 'bias'].unsqueeze(0).unsqueeze(3)).squeeze(1)


def coupling(p, g, st, prefix, warp, mode):
    keep = [d for d in (0, 1, 2) if d not in warp]
    eps = st[prefix + 'eps']
    pk = p[:, keep, :].contiguous()
    logvar, mu = torch.zeros_like(p), torch.zeros_like(p)
    logvar[:, list(warp), :] = F.softsign(_branch(pk, g, st, prefix, 'logvar', eps))
    mu[:, list(warp), :] = _branch(pk, g, st, prefix, 'mu', eps)
    scale = torch.sqrt(torch.add(eps, torch.exp(logvar)))
    return (scale * p + mu if mode == 'direct' else (p - mu) / scale), mu, logvar


def decoder_fused(p, g, st, n_flows, mode, grad=False, training=False):
    """(B,3,N), (B,G) torch CPU tensors -> (final coordinates, sum of logvars).
    grad=True keeps the autograd graph (tests use it as the gradient reference); training=True uses batch
    statistics and updates the running statistics in `st` in place (nn.BatchNorm1d semantics)."""
    C = 3 * n_flows
    order = range(C) if mode == 'direct' else range(C - 1, -1, -1)
    cur, logdet = p, None
    _TRAIN[0] = training
    try:
        with torch.set_grad_enabled(grad):
            for c in order:
                prefix = f'flows.{c // 3}.nvp{c % 3 + 1}.'
                cur, _, lv = coupling(cur, g, st, prefix, PATTERNS[c % 6], mode)
                logdet = lv if logdet is None else logdet + lv
    finally:
        _TRAIN[0] = False
    return cur, logdet
