"""CPU oracle for the discrete point-flow hot path.  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the arithmetic the reference performs in
``lib/networks/{layers,flows,decoders,losses}.py``.  It is the *checker* for the
HIP path: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.  Nothing under ``go_with_the_flows_amd/``
imports it, and the product path raises when the HIP library is missing.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the genuine
reference in the build container and stores its inputs/outputs as ``.npz``
fixtures; ``tests/test_oracle_golden.py`` checks every function below against
them (fp32 and fp64).

All functions take a ``state`` mapping (the reference ``state_dict`` converted to
numpy arrays, same keys) and plain ndarrays.  ``dtype`` selects the arithmetic
type: ``np.float32`` mirrors the reference bit-for-tolerance, ``np.float64`` is
the "ground truth" used to size tolerances.

Layout everywhere: points ``(B, 3, N)`` channel-major, latent ``(B, G)``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Mapping, Sequence, Tuple

import numpy as np

BN_EPS = 1e-5       # torch.nn.BatchNorm1d default (reference flows.py:27 uses defaults)
BN_MOMENTUM = 0.1   # idem

# warp patterns of one Triple: reference flows.py:129-148
TRIPLE_PATTERNS = {
    0: ([0], [1], [2]),
    1: ([0, 1], [0, 2], [1, 2]),
}


def keep_of(warp: Sequence[int]) -> List[int]:
    """Coordinates NOT warped by a coupling (reference flows.py:19-23)."""
    return [d for d in (0, 1, 2) if d not in warp]


# --------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------
def shared_dot(weight: np.ndarray, x: np.ndarray, bias: np.ndarray | None = None) -> np.ndarray:
    """Per-point linear map, reference layers.py:40-45.

    weight (1,out,in), x (B,in,N) -> (B,out,N); optional bias (1,out)."""
    y = np.einsum('oi,bin->bon', weight[0], x, optimize=True).astype(x.dtype, copy=False)
    if bias is not None:
        y = y + bias[0][None, :, None]
    return y


def batch_norm(x: np.ndarray, st: Mapping[str, np.ndarray], prefix: str, training: bool,
               affine: bool, new_stats: Dict[str, np.ndarray] | None = None) -> np.ndarray:
    """nn.BatchNorm1d over channel dim 1 of (B,f) or (B,f,N); reference flows.py:27,30,35,42.

    eval: running statistics.  train: biased batch variance for the
    normalisation, unbiased for the running update (torch semantics);
    updated statistics are written into ``new_stats`` if given."""
    dt = x.dtype
    axes = (0,) if x.ndim == 2 else (0, 2)
    shape = (1, -1) if x.ndim == 2 else (1, -1, 1)
    if training:
        n = x.size // x.shape[1]
        mean = x.mean(axis=axes, dtype=dt)
        var = ((x - mean.reshape(shape)) ** 2).mean(axis=axes, dtype=dt)
        if new_stats is not None:
            unbiased = var * dt.type(n / max(n - 1, 1))
            rm = st[prefix + 'running_mean'].astype(dt)
            rv = st[prefix + 'running_var'].astype(dt)
            new_stats[prefix + 'running_mean'] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean
            new_stats[prefix + 'running_var'] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * unbiased
            new_stats[prefix + 'num_batches_tracked'] = st[prefix + 'num_batches_tracked'] + 1
    else:
        mean = st[prefix + 'running_mean'].astype(dt)
        var = st[prefix + 'running_var'].astype(dt)
    y = (x - mean.reshape(shape)) / np.sqrt(var.reshape(shape) + dt.type(BN_EPS))
    if affine:
        y = y * st[prefix + 'weight'].astype(dt).reshape(shape) + st[prefix + 'bias'].astype(dt).reshape(shape)
    return y.astype(dt, copy=False)


def swish(x: np.ndarray) -> np.ndarray:
    """x * sigmoid(x), reference layers.py:9-10."""
    return x / (1 + np.exp(-x))


def softsign(x: np.ndarray) -> np.ndarray:
    return x / (1 + np.abs(x))


# --------------------------------------------------------------------------
# one elementary coupling: reference flows.py:95-117
# --------------------------------------------------------------------------
def film_mlp(g: np.ndarray, st, prefix: str, X: str, which: str, training: bool, new_stats=None) -> np.ndarray:
    """T_X_0_cond_{w|b}: Linear(G->f, no bias) -> BN -> Swish -> Linear(f->f);
    reference flows.py:33-45 / 68-80.  g (B,G) -> (B,f)."""
    base = f'{prefix}T_{X}_0_cond_{which}.{X}_sd1_film_{which}'
    h = g @ st[base + '0.weight'].astype(g.dtype).T
    h = batch_norm(h, st, base + '0_bn.', training, True, new_stats)
    h = swish(h)
    return (h @ st[base + '1.weight'].astype(g.dtype).T + st[base + '1.bias'].astype(g.dtype)).astype(g.dtype, copy=False)


def branch(p_keep: np.ndarray, g: np.ndarray, st, prefix: str, X: str, training: bool, eps, new_stats=None):
    """One branch (X='mu' or 'logvar') up to and including sd2(+bias); reference flows.py:99-107."""
    dt = p_keep.dtype
    t0 = f'{prefix}T_{X}_0.{X}_'
    h = shared_dot(st[t0 + 'sd0.weight'].astype(dt), p_keep)
    h = batch_norm(h, st, t0 + 'sd0_bn.', training, True, new_stats)
    h = np.maximum(h, 0)
    h = shared_dot(st[t0 + 'sd1.weight'].astype(dt), h)
    h = batch_norm(h, st, t0 + 'sd1_bn.', training, False, new_stats)
    a = eps + np.exp(film_mlp(g, st, prefix, X, 'w', training, new_stats))[:, :, None]
    b = film_mlp(g, st, prefix, X, 'b', training, new_stats)[:, :, None]
    h = np.maximum(a * h + b, 0)
    t1 = f'{prefix}T_{X}_1.{X}_sd2.'
    return shared_dot(st[t1 + 'weight'].astype(dt), h, st[t1 + 'bias'].astype(dt))


def coupling_forward(p: np.ndarray, g: np.ndarray, st, prefix: str, warp: Sequence[int], mode: str,
                     training: bool = False, new_stats=None) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """CondRealNVPFlow3D.forward, reference flows.py:95-117.

    Quirks reproduced on purpose: the affine is applied
    to all three coordinates, so kept ones are scaled by sqrt(eps+1); mu/logvar
    are exactly zero on kept coordinates."""
    dt = p.dtype
    keep = keep_of(warp)
    eps = st[prefix + 'eps'].astype(dt)[0]
    p_keep = np.ascontiguousarray(p[:, keep, :])
    logvar = np.zeros_like(p)
    mu = np.zeros_like(p)
    logvar[:, warp, :] = softsign(branch(p_keep, g, st, prefix, 'logvar', training, eps, new_stats))
    mu[:, warp, :] = branch(p_keep, g, st, prefix, 'mu', training, eps, new_stats)
    scale = np.sqrt(eps + np.exp(logvar))
    if mode == 'direct':
        p_out = scale * p + mu
    elif mode == 'inverse':
        p_out = (p - mu) / scale
    else:
        raise ValueError(mode)
    return p_out.astype(dt, copy=False), mu, logvar


def triple_forward(p, g, st, prefix: str, pattern: int, mode: str, training=False, new_stats=None):
    """CondRealNVPFlow3DTriple.forward, reference flows.py:150-160.  Lists are
    returned in nvp1..nvp3 order for both modes."""
    warps = TRIPLE_PATTERNS[pattern]
    order = (0, 1, 2) if mode == 'direct' else (2, 1, 0)
    ps, mus, lvs = [None] * 3, [None] * 3, [None] * 3
    cur = p
    for j in order:
        cur, mus[j], lvs[j] = coupling_forward(cur, g, st, f'{prefix}nvp{j + 1}.', warps[j], mode, training, new_stats)
        ps[j] = cur
    return ps, mus, lvs


def decoder_forward(p, g, st, n_flows: int, mode: str = 'direct', prefix: str = '', training=False, new_stats=None):
    """LocalCondRNVPDecoder.forward, reference decoders.py:61-79.

    Returns direct-ordered lists (ps, mus, logvars) of 3*n_flows arrays:
    ``ps[0]`` is base-space z after a full inverse, ``ps[-1]`` data-space x
    after a full direct."""
    ps, mus, lvs = [], [], []
    for i in range(n_flows):
        if mode == 'direct':
            cur = p if i == 0 else ps[-1]
            a, b, c = triple_forward(cur, g, st, f'{prefix}flows.{i}.', i % 2, mode, training, new_stats)
            ps, mus, lvs = ps + a, mus + b, lvs + c
        elif mode == 'inverse':
            cur = p if i == 0 else ps[0]
            t = n_flows - 1 - i
            a, b, c = triple_forward(cur, g, st, f'{prefix}flows.{t}.', t % 2, mode, training, new_stats)
            ps, mus, lvs = a + ps, b + mus, c + lvs
        else:
            raise ValueError(mode)
    return ps, mus, lvs


def decoder_fused(p, g, st, n_flows: int, mode: str, prefix: str = '', training=False):
    """What the consumers actually read (last paragraph): final
    coordinates and the per-dim sum of all logvars."""
    ps, _, lvs = decoder_forward(p, g, st, n_flows, mode, prefix, training)
    out = ps[0] if mode == 'inverse' else ps[-1]
    return out, sum(lvs)


def param_count(n_flows: int, f: int, G: int) -> int:
    """LocalCondRNVPDecoder.get_param_count, reference decoders.py:54-59 (the
    sizing formula, not the true count)."""
    return n_flows * 3 * (18 * f + 4 * f * G + 6 * f * f)


# --------------------------------------------------------------------------
# log-det accumulation + Gaussian base + mixture reduction: reference losses.py
# --------------------------------------------------------------------------
def point_flow_nll(samples0, mu0, logvar0, logvars: Sequence[np.ndarray]) -> np.ndarray:
    """PointFlowNLL.forward, reference losses.py:11-20.  ``logvars`` is the
    complete list INCLUDING the base logvar0 at index 0.  Returns (B,1,N)."""
    logdet = sum(logvars)
    q = logdet + (samples0 - mu0) ** 2 / np.exp(logvar0)
    return 0.5 * (q.sum(axis=1, keepdims=True) + math.log(2.0 * math.pi) * samples0.shape[1])


def flow_mixture_nll(components: Sequence[Mapping[str, Sequence[np.ndarray]]], logits: np.ndarray):
    """FlowMixtureNLL.forward, reference losses.py:88-137.

    ``components[k]`` holds the lists 'p_prior_samples', 'p_prior_mus',
    'p_prior_logvars' exactly as ``one_flow_decode`` builds them (reference
    models.py:195-205).  Returns the scalar loss and the (B,) per-shape NLLs."""
    dt = logits.dtype
    m = logits.max(axis=-1, keepdims=True)
    lse = m + np.log(np.exp(logits - m).sum(axis=-1, keepdims=True))
    log_w = np.log(np.exp(logits)) - lse                                   # (B,K)
    B = components[0]['p_prior_mus'][0].shape[0]
    per_shape = []
    for i in range(B):
        cols = []
        for comp in components:
            mu0 = comp['p_prior_mus'][0][i]
            lv0 = comp['p_prior_logvars'][0][i]
            logdet = sum(comp['p_prior_logvars'])[i]
            z = comp['p_prior_samples'][0][i]
            part1 = -(logdet + (z - mu0) ** 2 / np.exp(lv0)).sum(axis=0, keepdims=True)
            part2 = -math.log(2.0 * math.pi) * z.shape[0]
            cols.append(0.5 * (part1 + part2))
        lp = np.concatenate(cols, axis=0).T + log_w[i][None, :]            # (N,K)
        mm = lp.max(axis=-1, keepdims=True)
        lse_k = (mm + np.log(np.exp(lp - mm).sum(axis=-1, keepdims=True)))[:, 0]
        per_shape.append(-lse_k.sum(dtype=dt))
    per_shape = np.asarray(per_shape, dtype=dt)
    return per_shape.mean(dtype=dt), per_shape


def mixture_nll_fused(z: np.ndarray, logdet: np.ndarray, mu0: np.ndarray, lv0: np.ndarray, logits: np.ndarray):
    """Same value as ``flow_mixture_nll`` from the fused quantities.

    z, logdet: (K,B,3,N) final inverse coordinates and sum of the C coupling
    logvars (WITHOUT the base); mu0, lv0: (K,B,3) base Gaussian; logits (B,K)."""
    K, B = z.shape[:2]
    comps = []
    for k in range(K):
        m0 = np.broadcast_to(mu0[k][:, :, None], z[k].shape)
        l0 = np.broadcast_to(lv0[k][:, :, None], z[k].shape)
        comps.append({'p_prior_samples': [z[k]], 'p_prior_mus': [m0], 'p_prior_logvars': [l0, logdet[k]]})
    return flow_mixture_nll(comps, logits)


def to_numpy_state(state_dict) -> Dict[str, np.ndarray]:
    """torch state_dict -> numpy mapping (keys unchanged)."""
    return {k: (v.detach().cpu().numpy() if hasattr(v, 'detach') else np.asarray(v)) for k, v in state_dict.items()}


def cast_inputs(dtype, *arrays):
    return tuple(np.asarray(a).astype(dtype) for a in arrays)


# --------------------------------------------------------------------------
# optimiser: reference lib/networks/optimizers.py:15-76
# --------------------------------------------------------------------------
def adam_step(p, g, m, v, vmax, step, lr, beta1, beta2, eps, weight_decay, amsgrad):
    """One update of the reference's Adam/AMSGrad on numpy arrays (returns new p, m, v, vmax); `step` is 1-based."""
    dt = p.dtype.type
    m = m * dt(beta1) + dt(1 - beta1) * g
    v = v * dt(beta2) + dt(1 - beta2) * g * g
    if amsgrad:
        vmax = np.maximum(vmax, v)
        denom = np.sqrt(vmax)
    else:
        denom = np.sqrt(v)
    bc1 = dt(1 - beta1 ** step)
    bc2 = dt(math.sqrt(1 - beta2 ** step))
    upd = (m / bc1) / (denom / bc2 + dt(eps))
    if weight_decay != 0:
        p = p - (p * dt(weight_decay) + dt(lr) * upd)
    else:
        p = p - dt(lr) * upd
    return p, m, v, vmax
