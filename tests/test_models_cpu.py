"""Host logic of the model mirrors: checkpoint contract of the full model, decoder sizing rules, loud failure on CPU."""
import json
import os

import pytest
import torch

from conftest import GOLDEN
from go_with_the_flows_amd import models, _lib


def _cfg(**over):
    spec = json.load(open(os.path.join(GOLDEN, 'contract_model.json')))
    return dict(spec['cfg'], **over), spec['state_dict']


@pytest.mark.parametrize('base_type', ['free', 'freevar'])
def test_full_model_state_dict_matches_reference(base_type):
    cfg, ref_keys = _cfg(p_decoder_base_type=base_type)
    sd = models.Flow_Mixture_Model(**cfg).state_dict()
    if base_type == 'free':          # the stored contract is the generating/free model
        assert [k for k, _, _ in ref_keys] == list(sd.keys())
        for k, shape, dtype in ref_keys:
            assert list(sd[k].shape) == shape and str(sd[k].dtype) == 'torch.' + dtype, k
    else:
        assert 'p_prior_mus' in sd and 'p_prior.logvars.logvar_mlp0.weight' not in sd


@pytest.mark.parametrize('G,K,expect', [(128, 4, (11, 37)), (512, 4, (11, 33)), (128, 16, (6, 19)), (128, 1, (21, 64))])
def test_decoder_sizing_rules_resolve_to_the_surveyed_dimensions(G, K, expect):
    """configs/*.yaml: p_decoder_n_flows=21, p_decoder_n_features=64, depth_and_feature."""
    cfg, _ = _cfg(g_latent_space_size=G, n_components=K, p_decoder_n_flows=21, p_decoder_n_features=64,
                  params_reduce_mode='depth_and_feature')
    m = models.Flow_Mixture_Model.__new__(models.Flow_Mixture_Model)
    for k in ('n_components', 'params_reduce_mode', 'p_decoder_n_flows', 'p_decoder_n_features', 'g_latent_space_size'):
        object.__setattr__(m, k, cfg[k])
    assert m._get_decoder_params() == expect
    for mode in ('depth_first', 'feature_first'):
        object.__setattr__(m, 'params_reduce_mode', mode)
        depth, f = m._get_decoder_params()
        assert 1 <= depth <= 21 and 4 <= f <= 64
    object.__setattr__(m, 'params_reduce_mode', 'bogus')
    if K > 1:
        with pytest.raises(ValueError):
            m._get_decoder_params()


def test_model_refuses_cpu_tensors():
    cfg, _ = _cfg()
    m = models.Flow_Mixture_Model(**cfg).eval()
    with pytest.raises(_lib.GwtfError):
        m(torch.zeros(2, 3, 8), torch.zeros(2, 3, 8))
    with pytest.raises(ValueError):
        models.Flow_Mixture_Model(**dict(cfg, util_mode='generating')).forward_fused(torch.zeros(2, 3, 8), torch.zeros(2, 3, 8))


def test_loss_shortcuts_return_the_underlying_tensor_only_when_it_is_the_same_thing():
    """models._restack / _first_column (what keeps the loss gradient on dense routes): the (K, ...) tensor behind its K in-order
    slices, the (B, P) tensor behind its broadcast -- and a plain stack / column whenever anything differs; gradients agree."""
    from go_with_the_flows_amd.models import _first_column, _restack
    base = (torch.arange(24.0).view(3, 2, 4) + 0.5).requires_grad_(True) * 1.0
    parts = list(base.unbind(0))
    assert _restack(parts) is base
    for other in ([parts[0], parts[2], parts[1]], parts[:2], [p.clone() for p in parts], [base[0], base[1], base[2][:, :4] * 1.0]):
        got = _restack(other)
        assert got is not base and torch.equal(got, torch.stack(other))
    detached = list(base.detach().unbind(0))
    assert _restack(detached)._base is None or _restack(detached).requires_grad is False
    wide = torch.arange(48.0).view(2, 3, 2, 4)                       # slices of a LARGER tensor's sub-block are not "all of base"
    assert torch.equal(_restack(list(wide[0].unbind(0))), wide[0])

    h = torch.randn(5, 3, requires_grad=True)
    hb = h * 2.0
    e = hb.unsqueeze(2).expand(5, 3, 7)
    assert _first_column(e) is hb
    for other in (e.contiguous(), hb.t().contiguous().t().unsqueeze(2).expand(5, 3, 7), torch.randn(5, 3, 7), hb[:, :2].unsqueeze(2).expand(5, 2, 7)):
        got = _first_column(other)
        assert got is not hb and torch.equal(got, other[:, :, 0])
    (g1,) = torch.autograd.grad((_first_column(e) ** 2).sum(), h, retain_graph=True)
    (g2,) = torch.autograd.grad((e[:, :, 0] ** 2).sum(), h)
    assert torch.equal(g1, g2)
