"""The FiLM conditioning heads in HIP (csrc/gwtf_film_train.hip, autograd.FilmHeadsFn) against the same heads written as torch ops in
float64 on the parameters' views of the raw arena (reference lib/networks/flows.py:33-45, 68-80, 100-106: Linear -> BatchNorm1d over
the latent rows -> Swish -> Linear; a = eps + exp(.)): outputs, BatchNorm statistics, and every gradient -- the parameters' (written in
place into the flat arena gradient) and the latent's.  Needs an MI355X."""
import numpy as np
import pytest
import torch

import go_with_the_flows_amd as gw
from go_with_the_flows_amd import _lib
from go_with_the_flows_amd import autograd as gwa
from go_with_the_flows_amd.flows import stacked_raw_arena
from go_with_the_flows_amd.synth import load_synth_

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
BN_EPS = 1e-5


def _torch_heads(raw, g_all, KC, f, G, row0, B, eps, training, dtype=torch.float64):
    """a, bsh (B, KC, 2, f), batch mean / biased var (KC, 2, 2, f) with plain torch ops on views of the arena."""
    P = gwa._gather_film(raw.to(dtype), KC, f, G)
    hg, hb, hrm, hrv = P['hbn']
    hraw = torch.einsum('bg,cxhfg->bcxhf', g_all.to(dtype), P['L0'])
    if training:
        mean, var = hraw.mean(0), hraw.var(0, unbiased=False)
    else:
        mean, var = hrm, hrv
    hbn = (hraw - mean) / torch.sqrt(var + BN_EPS) * hg + hb
    hn = hbn * torch.sigmoid(hbn)
    o = torch.einsum('bcxhi,cxhji->bcxhj', hn, P['L1']) + P['b1']
    a = eps + torch.exp(o[:, :, :, 0])
    return a[row0:row0 + B], o[row0:row0 + B, :, :, 1], mean, var


CASES = [  # K, L, f, G, B_all, row0, B
    (1, 1, 8, 16, 5, 0, 5),
    (2, 1, 19, 32, 7, 2, 3),
    (4, 11, 37, 128, 64, 0, 64),
    (1, 2, 33, 512, 16, 0, 16),
    (1, 1, 64, 128, 100, 30, 40),
    (1, 1, 96, 48, 9, 0, 9),
    # more than 128 latent rows (the gathered rows of a large data-parallel group): row blocks inside the kernels
    (1, 1, 19, 32, 129, 0, 129),
    (1, 2, 37, 128, 300, 120, 64),
    (2, 1, 33, 64, 512, 256, 256),
]


@pytest.mark.parametrize('training', [True, False])
@pytest.mark.parametrize('case', CASES, ids=lambda c: 'K%d_L%d_f%d_G%d_B%d_r%d_b%d' % c)
def test_film_heads_forward_and_gradients_match_float64_torch(case, training):
    K, L, f, G, Ball, row0, B = case
    decs = []
    for k in range(K):
        d = gw.LocalCondRNVPDecoder(L, f, G)
        load_synth_(d, 3 + k)
        decs.append(d.to(DEV))
    stack = gw.MixtureStack(decs)
    C, KC, eps = 3 * L, K * 3 * L, 1e-6
    rng = np.random.default_rng(5)
    g_all = torch.from_numpy(rng.standard_normal((Ball, G)).astype(np.float32)).to(DEV).requires_grad_(True)
    raw = stacked_raw_arena(stack.engines).detach().clone().requires_grad_(True)
    FP = _lib.lib().gwtf_padded_width(f)

    film_raw, mean, var = gwa.FilmHeadsFn.apply(raw, g_all, KC, f, G, row0, B, eps, training)
    assert film_raw.shape == (B, KC, 2, 2, FP)
    assert float(film_raw[..., f:].abs().max()) == 0.0 if FP > f else True
    w = torch.from_numpy(rng.standard_normal((B, KC, 2, 2, FP)).astype(np.float32)).to(DEV)
    (film_raw * w).sum().backward()
    g_raw_hip, g_g_hip = raw.grad.clone(), g_all.grad.clone()

    raw64 = raw.detach().double().requires_grad_(True)
    g64 = g_all.detach().double().requires_grad_(True)
    a, bsh, m_ref, v_ref = _torch_heads(raw64, g64, KC, f, G, row0, B, eps, training)
    ((a * w[:, :, :, 0, :f].double()).sum() + (bsh * w[:, :, :, 1, :f].double()).sum()).backward()

    rel = lambda x, y: float((x.double() - y).abs().max() / (y.abs().max() + 1e-30))
    assert rel(film_raw[:, :, :, 0, :f], a.detach()) < 2e-5 and rel(film_raw[:, :, :, 1, :f], bsh.detach()) < 2e-5
    assert rel(mean, m_ref.detach()) < 1e-5 and rel(var, v_ref.detach()) < 1e-5
    assert rel(g_g_hip, g64.grad) < 2e-4, rel(g_g_hip, g64.grad)
    # parameter gradients: compare slot by slot through the same views (running statistics get none in either)
    Ph, Pt = gwa._gather_film(g_raw_hip, KC, f, G), gwa._gather_film(raw64.grad, KC, f, G)
    for name in ('L0', 'L1', 'b1'):
        assert rel(Ph[name], Pt[name]) < 2e-4, (name, rel(Ph[name], Pt[name]))
    for i, name in enumerate(('gamma', 'beta')):
        assert rel(Ph['hbn'][i], Pt['hbn'][i]) < 2e-4, (name, rel(Ph['hbn'][i], Pt['hbn'][i]))
    # nothing outside the FiLM slots is written
    mask = torch.ones_like(g_raw_hip, dtype=torch.bool)
    probe = gwa._gather_film(torch.arange(g_raw_hip.numel(), device=DEV, dtype=torch.float64).view_as(g_raw_hip), KC, f, G)
    for t in (probe['L0'], probe['L1'], probe['b1'], probe['hbn'][0], probe['hbn'][1]):
        mask.view(-1)[t.reshape(-1).long()] = False
    assert float(g_raw_hip[mask].abs().max()) == 0.0


def test_film_heads_propagate_a_diverged_branch_as_nan_scale():
    K, L, f, G, B = 1, 1, 8, 16, 4
    d = gw.LocalCondRNVPDecoder(L, f, G)
    load_synth_(d, 3)
    stack = gw.MixtureStack([d.to(DEV)])
    raw = stacked_raw_arena(stack.engines).detach().clone()
    raw[0, 5] = float('nan')          # an sd0 weight of coupling 0, branch 0: nowhere near the FiLM heads
    g = torch.randn(B, G, device=DEV)
    film_raw, _, _ = gwa.FilmHeadsFn.apply(raw, g, 3 * L, f, G, 0, B, 1e-6, True)
    assert torch.isnan(film_raw[:, 0, 0, 0, :f]).all() and torch.isfinite(film_raw[:, 0, 1]).all()
    assert torch.isfinite(film_raw[:, 1:]).all()
