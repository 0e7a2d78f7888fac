"""Global prior flow: pin oracle/prior_oracle.py AND the torch modules to the genuine reference (tests/golden/g12_prior.npz)."""
import numpy as np
import pytest
import torch

from conftest import golden
from helpers import maxabs
from go_with_the_flows_amd import prior
from go_with_the_flows_amd.synth import load_synth_
from oracle import prior_oracle as po

TOL = 2e-5


def _case():
    G12 = golden('g12_prior')
    n_flows, F_, G, B = (int(v) for v in G12['dims'])
    m = prior.GlobalRNVPDecoder(n_flows, F_, G)
    st = load_synth_(m, 1210)
    return G12, m, st, n_flows, G, B


@pytest.mark.parametrize('training', [False, True])
@pytest.mark.parametrize('mode', ['direct', 'inverse'])
def test_oracle_and_module_match_reference(mode, training):
    G12, m, st, n_flows, G, B = _case()
    t = f'{"train" if training else "eval"}_{mode}'
    gs, mus, lvs = po.decoder(G12['g'], st, n_flows, mode, training)
    for got, key in ((gs, 'gs_'), (mus, 'mus_'), (lvs, 'lvs_')):
        assert maxabs(np.stack(got), G12[key + t]) < TOL
    m.train(training)
    with torch.no_grad():
        tgs, tmus, tlvs = m(torch.from_numpy(G12['g']), mode=mode)
    assert len(tgs) == len(tmus) == len(tlvs) == 2 * n_flows
    for got, key in ((tgs, 'gs_'), (tmus, 'mus_'), (tlvs, 'lvs_')):
        assert maxabs(torch.stack(got).numpy(), G12[key + t]) < TOL
    if training:
        sd = m.state_dict()
        assert maxabs(sd['flows.1.nvp2.T_mu_0.mu_mlp0_bn.running_mean'].numpy(), G12['rm_' + t]) < TOL
        assert maxabs(sd['flows.1.nvp2.T_mu_0.mu_mlp0_bn.running_var'].numpy(), G12['rv_' + t]) < TOL
    if mode == 'inverse':
        mu0, lv0 = np.broadcast_to(G12['mu0'], (B, G)), np.broadcast_to(G12['lv0'], (B, G))
        ref_nll = float(G12['gnll_' + t])
        assert abs(po.gaussian_flow_nll(gs + [G12['g'].astype(np.float64)], [mu0] + mus, [lv0] + lvs) - ref_nll) < 1e-5 * abs(ref_nll)
        nll = prior.GaussianFlowNLL()(tgs + [torch.from_numpy(G12['g'])], [torch.from_numpy(mu0.copy())] + tmus,
                                      [torch.from_numpy(lv0.copy())] + tlvs)
        assert abs(float(nll) - ref_nll) < 1e-5 * abs(ref_nll)
        ent = float(G12['gent_' + t])
        assert abs(po.gaussian_entropy(G12['post_lv'].astype(np.float64)) - ent) < 1e-5 * abs(ent)
        assert abs(float(prior.GaussianEntropy()(torch.from_numpy(G12['post_lv']))) - ent) < 1e-5 * abs(ent)


def test_round_trip_and_gradients():
    _, m, _, n_flows, G, B = _case()
    m.eval()
    g = torch.randn(B, G, dtype=torch.float32)
    with torch.no_grad():
        z = m(g, mode='inverse')[0][0]
        back = m(z, mode='direct')[0][-1]
    assert maxabs(back.numpy(), g.numpy()) < 1e-4
    g.requires_grad_()
    gs, mus, lvs = m(g, mode='inverse')
    (gs[0].square().sum() + sum(lvs).sum()).backward()
    assert torch.isfinite(g.grad).all() and all(p.grad is not None for p in m.parameters())


def test_general_warp_indices_fall_back_to_index_gather():
    f = prior.RealNVPFlow(8, 6, warp_inds=[0, 1, 4])
    assert f.keep_inds == [2, 3, 5] and f._warp_sl is None and f._keep_sl is None
    g = torch.randn(3, 6)
    out, mu, lv = f(g, mode='direct')
    assert torch.equal(mu[:, f.keep_inds], torch.zeros(3, 3)) and torch.equal(out[:, f.keep_inds], g[:, f.keep_inds])
