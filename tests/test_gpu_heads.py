"""The per-shape heads (FeatureEncoder / WeightsEncoder, reference lib/networks/encoders.py:31-91) on the HIP path
(csrc/gwtf_heads.hip: one launch per layer, forward and backward) against the GENUINE reference's outputs and gradients (golden g20:
g_posterior, p_prior and mixture_weights_encoder at the shipped configs' sizes, train and eval BatchNorm, the reference's fp64 run with
its own fp32 noise recorded per tensor).  Needs an MI355X."""
import numpy as np
import pytest
import torch

from conftest import golden
from go_with_the_flows_amd import encoders
from go_with_the_flows_amd.synth import synth_state

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
CASES = {'post': (encoders.FeatureEncoder, dict(deterministic=False), 2000),
         'pprior': (encoders.FeatureEncoder, dict(deterministic=False), 2010),
         'wts': (encoders.WeightsEncoder, dict(deterministic=True), 2020)}


def build(tag, D):
    cls, kw, seed = CASES[tag]
    m = cls(*[int(v) for v in D[f'{tag}_ctor']], **kw)
    st = synth_state(m.state_dict(), seed)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    return m


def rel(a, b):
    b = np.asarray(b, np.float64)
    return float(np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize('training', [True, False])
@pytest.mark.parametrize('tag', sorted(CASES))
def test_heads_match_the_reference_outputs_gradients_and_buffers(tag, training):
    D = golden('g20_heads')
    m = build(tag, D).to(DEV).train(training)
    m.hip_max_width = 4096                                     # (the default) every head on the HIP kernels
    t = 'train' if training else 'eval'
    x = torch.from_numpy(D[f'{tag}_x']).to(DEV).requires_grad_(True)
    assert m._hip_layers(x) is not None                        # this call is served by the HIP kernels, not by library modules
    y = m(x)
    ys = y if isinstance(y, tuple) else (y,)
    wgt = torch.from_numpy(D[f'{tag}_wgt']).to(DEV)
    sum((o * wgt[i]).sum() for i, o in enumerate(ys)).backward()
    got = {f'out{i}': o for i, o in enumerate(ys)}
    got['gx'] = x.grad
    got.update({f'grad.{n}': p.grad for n, p in m.named_parameters()})
    worst = {}
    for key, v in got.items():
        want, noise = D[f'{tag}_{t}_{key}'], float(D[f'{tag}_{t}_noise.{key}'])
        v = v.detach().cpu().numpy()
        if v.size > 65536:
            v = v[::8]                                        # the fixture keeps every 8th row of the large weight gradients
        worst[key] = (rel(v, want), noise)
        # no further from the reference's fp64 run than 4x its own fp32 run (+ 1e-6 of the tensor's largest entry)
        assert worst[key][0] < 4 * noise + 1e-6, (tag, t, key, worst[key])
    if training:
        for name, buf in m.named_buffers():
            want = D[f'{tag}_buf.{name}']
            assert rel(buf.detach().cpu().numpy().astype(np.float64), want.astype(np.float64)) < 1e-5, name


@pytest.mark.parametrize('B,din,dout,layers', [(2, 37, 5, 2), (128, 130, 70, 1), (17, 512, 128, 1), (64, 64, 64, 3),
                                                 (129, 40, 33, 2), (300, 512, 128, 1), (512, 128, 6, 3)])
def test_heads_odd_sizes_against_the_library_modules_in_fp64(B, din, dout, layers):
    """Row counts 2 .. 512 (beyond 128: the kernels' row blocks), widths around the 16-column workgroup blocks and the split-K threshold, no-BatchNorm trunks, the
    closed-form replay of `bn_updates` running-statistic updates: HIP path == the same module evaluated by torch on the CPU in fp64."""
    for batch_norm in (True, False):
        torch.manual_seed(B + din)
        ref = encoders.FeatureEncoder(layers, din, dout, deterministic=False, batch_norm=batch_norm, easy_init=True).double().train()
        m = encoders.FeatureEncoder(layers, din, dout, deterministic=False, batch_norm=batch_norm, easy_init=True)
        m.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in ref.state_dict().items()})
        m = m.to(DEV).train()
        m.hip_max_width = 4096
        x64 = torch.randn(B, din, dtype=torch.float64)
        w = torch.randn(2, B, dout, dtype=torch.float64)
        xr = x64.clone().requires_grad_(True)
        for _ in range(3):                                    # three evaluations of the same batch = bn_updates=3
            ref.zero_grad()
            xr.grad = None
            mu_r, lv_r = ref(xr)
            ((mu_r * w[0]).sum() + (lv_r * lv_r * w[1]).sum()).backward()
        xd = x64.float().to(DEV).requires_grad_(True)
        assert m._hip_layers(xd) is not None
        mu, lv = m(xd, bn_updates=3)
        ((mu * w[0].float().to(DEV)).sum() + (lv * lv * w[1].float().to(DEV)).sum()).backward()
        assert rel(mu.detach().cpu().numpy(), mu_r.detach().numpy()) < 2e-5 and rel(lv.detach().cpu().numpy(), lv_r.detach().numpy()) < 2e-5
        assert rel(xd.grad.cpu().numpy(), xr.grad.numpy()) < 5e-5
        for (n, p), pr in zip(m.named_parameters(), ref.parameters()):
            assert rel(p.grad.cpu().numpy(), pr.grad.numpy()) < 5e-5, (n, batch_norm)
        for (n, b), br in zip(m.named_buffers(), ref.buffers()):
            assert rel(b.cpu().numpy().astype(np.float64), br.numpy().astype(np.float64)) < 1e-5, (n, batch_norm)


def test_heads_refuse_what_the_kernels_do_not_cover_and_fall_back_to_the_library_modules():
    m = encoders.FeatureEncoder(1, 16, 4).to(DEV).train()
    assert m._hip_layers(torch.randn(129, 16, device=DEV)) is not None      # any number of rows is served by the kernels
    assert m._hip_layers(torch.randn(8, 16, device=DEV, dtype=torch.float64)) is None   # other dtypes: torch modules on the device
    assert m.double()(torch.randn(8, 16, device=DEV, dtype=torch.float64))[0].shape == (8, 4)
    m = m.float()
    with pytest.raises(ValueError):                            # one row in train mode: BatchNorm raises (torch raises the same)
        m(torch.randn(1, 16, device=DEV))


@pytest.mark.parametrize('B,din,dout', [(64, 512, 128), (64, 128, 3), (130, 40, 33)])
def test_head_pair_launch_equals_the_two_single_launches_bit_for_bit(B, din, dout, monkeypatch):
    """The mu / logvar heads in one launch (csrc/gwtf_heads.hip head_*_pair_kernel): same blocks, same products as one launch per head;
    also with only ONE of the two outputs in the loss (the other head's upstream gradient is None)."""
    torch.manual_seed(B + dout)
    m = encoders.FeatureEncoder(1, din, dout, deterministic=False, easy_init=True).to(DEV).train()
    x0 = torch.randn(B, din, device=DEV)
    w = torch.randn(2, B, dout, device=DEV)
    res = {}
    for use_b in (True, False):
        for name in ('pair', 'single'):
            if name == 'single':
                monkeypatch.setenv('GWTF_NO_HEAD_PAIR', '1')
            else:
                monkeypatch.delenv('GWTF_NO_HEAD_PAIR', raising=False)
            m.zero_grad(set_to_none=True)
            st = {k: v.clone() for k, v in m.state_dict().items()}
            x = x0.clone().requires_grad_(True)
            mu, lv = m(x)
            ((mu * w[0]).sum() + ((lv * lv * w[1]).sum() if use_b else 0.0)).backward()
            res[name] = [mu, lv, x.grad] + [p.grad for p in m.parameters()]
            m.load_state_dict(st)                                  # (the trunk's running statistics moved)
        for i, (a, b) in enumerate(zip(res['pair'], res['single'])):
            if i == 2 and use_b:
                # dL/dx: the pair's second product accumulates onto the first inside the kernel, two launches are summed by autograd --
                # the same terms in another association
                assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())
            else:
                assert (a is None and b is None) or torch.equal(a, b), i
