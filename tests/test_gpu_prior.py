"""The global prior flow as ONE HIP launch per direction (csrc/gwtf_prior.hip) against the genuine reference (golden
g12_prior: GlobalRNVPDecoder of lib/networks/decoders.py:7-38, eval and train BatchNorm, both modes, updated running
statistics, GaussianFlowNLL) and, for gradients, against CPU autograd of the module-by-module torch evaluation (itself
pinned to the reference by tests/test_prior_cpu.py).  Needs an MI355X."""
import numpy as np
import pytest
import torch

from conftest import golden
from helpers import maxabs
from go_with_the_flows_amd import prior
from go_with_the_flows_amd.synth import load_synth_

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TOL = 2e-5


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def host(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize('training', [False, True])
@pytest.mark.parametrize('mode', ['direct', 'inverse'])
def test_fused_prior_flow_matches_reference(mode, training):
    G12 = golden('g12_prior')
    n_flows, F_, G, B = (int(v) for v in G12['dims'])
    m = prior.GlobalRNVPDecoder(n_flows, F_, G)
    load_synth_(m, 1210)
    m = m.to(DEV).train(training)
    assert m._fused_ok(dev(G12['g']))                       # the HIP path is the one that runs
    t = f'{"train" if training else "eval"}_{mode}'
    with torch.no_grad():
        gs, mus, lvs = m(dev(G12['g']), mode=mode)
    assert len(gs) == len(mus) == len(lvs) == 2 * n_flows
    for got, key in ((gs, 'gs_'), (mus, 'mus_'), (lvs, 'lvs_')):
        assert maxabs(host(torch.stack(got)), G12[key + t]) < TOL, key
    if training:
        sd = m.state_dict()
        assert maxabs(host(sd['flows.1.nvp2.T_mu_0.mu_mlp0_bn.running_mean']), G12['rm_' + t]) < TOL
        assert maxabs(host(sd['flows.1.nvp2.T_mu_0.mu_mlp0_bn.running_var']), G12['rv_' + t]) < TOL
        assert int(sd['flows.0.nvp1.T_logvar_0.logvar_mlp0_bn.num_batches_tracked']) == 1
    if mode == 'inverse':
        mu0, lv0 = dev(np.broadcast_to(G12['mu0'], (B, G)).copy()), dev(np.broadcast_to(G12['lv0'], (B, G)).copy())
        nll = prior.GaussianFlowNLL()(gs + [dev(G12['g'])], [mu0] + mus, [lv0] + lvs)
        assert abs(float(nll) - float(G12['gnll_' + t])) < 1e-5 * abs(float(G12['gnll_' + t]))


@pytest.mark.parametrize('cfg', [(3, 24, 16, 6), (7, 128, 128, 64), (2, 40, 35, 5), (7, 128, 512, 16),
                                 (2, 40, 35, 130), (2, 128, 128, 300), (1, 64, 512, 512)])
@pytest.mark.parametrize('training', [False, True])
@pytest.mark.parametrize('mode', ['direct', 'inverse'])
def test_fused_prior_flow_gradients(cfg, mode, training):
    """Every parameter, the input, and list slots in the middle of the stack: HIP backward == CPU autograd of the torch
    modules.  (7,128,128,64) / (7,128,512,16) are the shipped configs' prior flows (airplane; autoencoding per GPU),
    (2,40,35,5) has an odd latent width and a hidden width that is no multiple of 16; the cases with 130 / 300 / 512 rows (the gathered
    rows of a large data-parallel group) walk the batch in row blocks inside the kernels."""
    assert prior.GlobalRNVPDecoder(cfg[0], cfg[1], cfg[2]).to(DEV)._fused_ok(torch.zeros(cfg[3], cfg[2], device=DEV))
    n_flows, F_, G, B = cfg
    ref = prior.GlobalRNVPDecoder(n_flows, F_, G)
    load_synth_(ref, 77)
    if n_flows > 3:          # the synthetic output layers (std 0.08) make a 14-flow stack overflow even in fp64: tame them
        with torch.no_grad():
            for k, v in ref.named_parameters():
                if 'mlp1' in k:
                    v.mul_(0.15)
    m = prior.GlobalRNVPDecoder(n_flows, F_, G)
    m.load_state_dict(ref.state_dict())
    ref = ref.double().train(training)
    m = m.to(DEV).train(training)
    rng = np.random.default_rng(5)
    g = rng.standard_normal((B, G)).astype(np.float32)
    n2 = 2 * n_flows
    w_gs = rng.standard_normal((n2, B, G)).astype(np.float32) * (rng.random((n2, 1, 1)) < 0.5)
    w_lv = rng.standard_normal((n2, B, G)).astype(np.float32)
    gd = dev(g).requires_grad_(True)
    gs, mus, lvs = m(gd, mode=mode)
    loss = sum((gs[j] * dev(w_gs[j])).sum() + (lvs[j] * dev(w_lv[j])).sum() for j in range(n2))
    loss.backward()
    gt = torch.from_numpy(g).double().requires_grad_(True)
    rgs, rmus, rlvs = ref(gt, mode=mode)
    rloss = sum((rgs[j] * torch.from_numpy(w_gs[j]).double()).sum() + (rlvs[j] * torch.from_numpy(w_lv[j]).double()).sum() for j in range(n2))
    rloss.backward()
    assert maxabs(host(torch.stack(gs)), torch.stack(rgs).detach().numpy()) < 1e-4
    named = dict(ref.named_parameters())
    gscale = max(float(v.grad.norm()) for v in named.values())        # parameters with an (analytically) zero gradient are
    rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / (np.linalg.norm(b) + 1e-4 * gscale))   # held to the global scale (fp32 noise of a cancelling sum)
    tol = 2e-3 if training else 2e-4          # batch statistics over few rows amplify fp32 rounding
    assert rel(host(gd.grad), gt.grad.numpy()) < tol
    worst = max((rel(host(v.grad), named[k].grad.numpy()), k, float(named[k].grad.norm())) for k, v in m.named_parameters())
    assert worst[0] < tol, (worst, gscale)
    with pytest.raises((NotImplementedError, RuntimeError)):      # a gradient through mus[j] is refused, not dropped
        gs, mus, lvs = m(dev(g).requires_grad_(True), mode=mode)
        (gs[0].sum() + mus[1].sum()).backward()


def test_full_model_uses_the_fused_prior_flow_and_matches_its_own_torch_path():
    """Flow_Mixture_Model.encode calls g_prior(..., mode='inverse') (models.py:137): same lists from the fused launch as
    from the module-by-module evaluation on the same device."""
    m = prior.GlobalRNVPDecoder(7, 128, 128)
    load_synth_(m, 9)
    with torch.no_grad():
        for k, v in m.named_parameters():
            if 'mlp1' in k:
                v.mul_(0.15)                  # keep the 14-flow stack inside the fp32 range (see above)
    m = m.to(DEV).eval()
    g = torch.randn(64, 128, device=DEV)
    with torch.no_grad():
        fused = m(g, mode='inverse')
        m._fused_ok = lambda _g, _rows=None: False
        plain = m(g, mode='inverse')
    for a, b in zip(fused, plain):
        assert maxabs(host(torch.stack(a)), host(torch.stack(b))) < 2e-5
