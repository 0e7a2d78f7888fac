"""Regression tests for the round-1 advisor findings: empty component clouds in evaluation-mode decode,
SyncBatchNorm-converted encoder, optimiser steps seen by the packed-weight caches, graph warm-up leaving no trace."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import GOLDEN, TOL_COORD, TOL_LOGDET
from helpers import decoder_and_state, maxabs
import go_with_the_flows_amd as gw
from go_with_the_flows_amd import models, optim, encoders
from go_with_the_flows_amd.synth import load_synth_, synth_inputs
from oracle import flow_oracle as fo

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def host(t):
    return t.detach().cpu().numpy()


def build(**over):
    cfg = dict(json.load(open(os.path.join(GOLDEN, 'contract_model.json')))['cfg'], **over)
    m = models.Flow_Mixture_Model(**cfg)
    load_synth_(m, 1310)
    return m.to(DEV), cfg


def test_empty_cloud_passes_through_every_entry_point():
    """A component that draws 0 points (reference flow_mixture.py:153-165 feeds a (1,3,0) cloud to its decoder)."""
    m, _ = decoder_and_state(2, 19, 16, 5)
    m = m.to(DEV).eval()
    p0, g = torch.zeros(1, 3, 0, device=DEV), torch.randn(1, 16, device=DEV)
    with torch.no_grad():
        for mode in ('direct', 'inverse'):
            ps, mus, lvs = m(p0, g, mode)
            assert len(ps) == len(mus) == len(lvs) == 6 and all(t.shape == (1, 3, 0) for t in ps + mus + lvs)
            out, ld = m.forward_fused(p0, g, mode)
            assert out.shape == ld.shape == (1, 3, 0)
    po, mu, lv = m.flows[0].nvp1(p0, g, 'direct')
    assert po.shape == (1, 3, 0)


def test_evaluation_decode_with_a_component_that_gets_no_point():
    m, cfg = build(util_mode='generating')
    m.eval()
    K = cfg['n_components']
    with torch.no_grad():
        m.mixture_weights_encoder.mus[-1].weight.zero_()
        m.mixture_weights_encoder.mus[-1].bias.copy_(torch.tensor([8.0, -30.0, 8.0][:K]))   # component 1: weight ~ e^-38
    rng = np.random.default_rng(3)
    gcloud, _ = synth_inputs(1, 40, 16, 7)
    np.random.seed(11)
    with torch.no_grad():
        enc, samples, labels, logits = m(dev(gcloud), dev(gcloud), None, 40, True, False)
    lab = host(labels)[0]
    assert (lab == 2).sum() == 0 and set(np.unique(lab)) <= {1.0, 3.0}      # labels are component index + 1
    assert samples.shape == (1, 3, 40) and np.isfinite(host(samples)).all()


def test_syncbatchnorm_converted_model_runs_the_fused_eval_encoder():
    """train_ae.py:152 converts EVERY BatchNorm of the model; the per-epoch eval() pass then runs the fused encoder."""
    m, _ = build()
    g, _ = synth_inputs(3, 64, 16, 21)
    m.eval()
    with torch.no_grad():
        before = m.pc_encoder(dev(g))
    m2 = nn.SyncBatchNorm.convert_sync_batchnorm(m)
    assert not any(isinstance(x, nn.BatchNorm1d) for x in m2.pc_encoder.modules())
    m2.eval()
    with torch.no_grad():
        after = m2.pc_encoder(dev(g))
    assert torch.equal(before, after)


def test_optimizer_step_invalidates_packed_weights_without_a_mode_toggle():
    """Frozen-BatchNorm fine-tuning: the module stays in eval(), a no-grad forward follows every optimiser step."""
    L, f, G, B, N = 1, 19, 16, 2, 50
    m, _ = decoder_and_state(L, f, G, 31)
    m = m.to(DEV).eval()
    p, g = synth_inputs(B, N, G, 32)
    pd, gd = dev(p), dev(g)
    opt = optim.Adam(m.parameters(), lr=5e-2)
    for it in range(3):                     # the second and third steps take the optimiser's cached-plan fast path
        with torch.no_grad():
            z0, _ = m.forward_fused(pd, gd, 'inverse')
        out, ld = m.forward_fused(pd, gd, 'inverse')
        (0.5 * (out ** 2).sum() + ld.sum()).backward()
        opt.step()
        opt.zero_grad()
        with torch.no_grad():
            z1, ld1 = m.forward_fused(pd, gd, 'inverse')
        st = {k: host(v) for k, v in m.state_dict().items()}
        ref, ref_ld = fo.decoder_fused(p, g, st, L, 'inverse')
        assert maxabs(host(z1), host(z0)) > 1e-3, it                  # the step was seen
        assert maxabs(host(z1), ref) < TOL_COORD and maxabs(host(ld1), ref_ld) < TOL_LOGDET, it


def test_graphed_train_step_construction_leaves_no_trace_in_the_model():
    from go_with_the_flows_amd.training import GraphedTrainStep
    m, cfg = build()
    m.train()
    loss_fn = models.Flow_Mixture_Loss(**cfg)
    opt = optim.Adam(m.parameters(), lr=1e-3)
    g, _ = synth_inputs(4, 48, 16, 41)
    p, _ = synth_inputs(4, 48, 16, 42)
    before = {k: v.detach().clone() for k, v in m.state_dict().items()}
    rng_before = torch.cuda.get_rng_state(torch.device(DEV)).clone()
    step = GraphedTrainStep(m, loss_fn, opt, dev(g), dev(p))
    torch.cuda.synchronize()
    after = m.state_dict()
    for k, v in before.items():
        assert torch.equal(v, after[k]), k
    assert torch.equal(rng_before, torch.cuda.get_rng_state(torch.device(DEV)))
    terms = step(dev(g), dev(p))
    assert np.isfinite(float(terms[0]))
    assert int(after['pc_decoder.0.flows.0.nvp1.T_mu_0.mu_sd0_bn.num_batches_tracked']) == 1
