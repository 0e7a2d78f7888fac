"""Fused latent-loss launch (csrc/gwtf_latent.hip) against the reference formulas in float64 torch (losses.py:24-41, :159-170)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LOG2PI = float(np.log(2.0 * np.pi))


def ref_terms(nll, z, mu0, lv0, flow_lv, post_lv, pw, gw, ew):
    B, G = z.shape
    lv_sum = lv0.unsqueeze(0) + flow_lv.sum(0)
    gnll = 0.5 * (torch.sum(lv_sum + (z - mu0) ** 2 / torch.exp(lv0)) / B + LOG2PI * G)
    gent = 0.5 * (G * (1.0 + LOG2PI) + post_lv.sum(1).mean())
    pnll = nll.mean()
    return torch.stack([pw * pnll + gw * gnll - ew * gent, pnll, gnll, gent])


@pytest.mark.parametrize('B,G,n2', [(64, 128, 28), (5, 7, 2), (1, 1, 1), (3, 512, 6), (130, 33, 4)])
def test_values_and_gradients_match_float64(B, G, n2):
    from go_with_the_flows_amd.prior import LatentLossFn
    gen = torch.Generator().manual_seed(B * 1000 + G)
    mk = lambda *s, scale=1.0: (torch.randn(*s, generator=gen) * scale)
    host = [mk(B) * 100 + 3000, mk(B, G), mk(G, scale=0.1), mk(G, scale=0.5), mk(n2, B, G, scale=0.3), mk(B, G, scale=0.7)]
    w = (1.0, 0.75, 0.3)
    dev = [t.cuda().requires_grad_(True) for t in host]
    f64 = [t.double().requires_grad_(True) for t in host]
    got, want = LatentLossFn.apply(*dev, *w), ref_terms(*f64, *w)
    assert torch.allclose(got.double().cpu(), want, rtol=2e-6, atol=1e-5), (got, want)
    up = torch.tensor([1.3, -0.2, 0.5, 0.9])
    got.backward(up.cuda())
    want.backward(up.double())
    for a, b, name in zip(dev, f64, ['nll', 'z', 'mu0', 'lv0', 'flow_lv', 'post_lv']):
        scale = max(1e-30, float(b.grad.abs().max()))
        assert float((a.grad.double().cpu() - b.grad).abs().max()) <= 5e-6 * scale, name


def test_loss_module_uses_the_fused_launch_and_matches_the_torch_combination(monkeypatch):
    from go_with_the_flows_amd.models import Flow_Mixture_Loss
    crit = Flow_Mixture_Loss(pnll_weight=1.0, gnll_weight=1.0, gent_weight=0.5, n_components=4)
    B, G, n2 = 6, 16, 4
    gen = torch.Generator().manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=gen).cuda()
    mu0, lv0 = r(1, G).requires_grad_(True), r(1, G).requires_grad_(True)
    stacked = r(n2, B, G).requires_grad_(True)
    z, post = r(B, G).requires_grad_(True), r(B, G).requires_grad_(True)
    nll = (r(B) + 50).requires_grad_(True)
    leaves = [mu0, lv0, stacked, z, post, nll]

    def prior():
        return {'g_prior_samples': [z, z], 'g_prior_mus': [mu0.expand(B, G)] + list(stacked.unbind(0)),
                'g_prior_logvars': [lv0.expand(B, G)] + list(stacked.unbind(0)), 'g_posterior_logvars': post,
                '_g_prior_logvars_stacked': stacked, '_g0_params': (mu0, lv0)}

    from go_with_the_flows_amd import prior as prior_mod
    calls = []
    orig = prior_mod.LatentLossFn.apply
    monkeypatch.setattr(prior_mod.LatentLossFn, 'apply', staticmethod(lambda *a: (calls.append(1), orig(*a))[1]))
    a = crit._combine(nll, prior())
    a[0].backward()
    ga = [t.grad.clone() for t in leaves]
    assert calls == [1]
    for t in leaves:
        t.grad = None
    monkeypatch.setenv('GWTF_NO_FUSED_LATENT_LOSS', '1')
    b = crit._combine(nll, prior())
    b[0].backward()
    assert calls == [1]
    for x, y in zip(a, b):
        assert abs(float(x) - float(y)) <= 2e-6 * max(1.0, abs(float(y)))
    for x, t in zip(ga, leaves):
        assert float((x - t.grad).abs().max()) <= 2e-6 * max(1e-30, float(t.grad.abs().max()))
