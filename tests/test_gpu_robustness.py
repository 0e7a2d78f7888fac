"""Robustness of the split-f16 path (parity holes c and d).

(c) adversarial ranges: per-feature weight scales over 10^+-3, BatchNorm gains up to 50, running variances from 1e-6 to
    1e+6, FiLM scales a down to ~1e-6, coordinates up to 50.  The reference is fp32 throughout (flows.py:95-117); the HIP
    path contracts sd1 on the f16 matrix unit with hi/lo-split operands, so its packer range-scales both operands with exact
    powers of two (csrc/gwtf_layout.h, RANGE SCALING).  Bar: the depth-scaled tolerance against the fp64 oracle AND no worse
    than 3x the fp32 oracle's own rounding noise.
(d) NaN / Inf in p, g or the weights must reach `out` / `logdet` (the reference aborts on a non-finite loss,
    training.py:43-46), and coordinates beyond the supported range (|x| > 3e4) must give NaN, never a silently wrong value.
"""
import numpy as np
import pytest
import torch

from conftest import TOL_COORD, TOL_LOGDET, tol_at_depth, record_parity
from helpers import decoder_and_state, state64, maxabs
import go_with_the_flows_amd as gw
from go_with_the_flows_amd import _lib
from go_with_the_flows_amd.synth import synth_inputs
from oracle import flow_oracle as fo

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def host(t):
    return t.detach().cpu().numpy()


def load(m, st):
    m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in st.items()})
    return m


def adversarial_state(st, p, g, L, seed, gain_max=50.0, decades=3.0, a_floor=-16.0):
    """Wild per-feature scales, then running statistics CALIBRATED to the activations they normalise (one train-mode pass
    of the fp64 oracle on the same inputs), as a trained checkpoint's would be: running variances end up spanning
    ~10^-6 .. 10^+6 while every BatchNorm output stays O(gain)."""
    rng = np.random.default_rng(seed)
    st = {k: v.copy() for k, v in st.items()}
    for k in st:
        v = st[k]
        if k.endswith('sd0.weight') or k.endswith('sd1.weight'):          # (1, out, in): rows and columns over 10^+-decades
            rows = 10.0 ** rng.uniform(-decades, decades, (1, v.shape[1], 1))
            cols = 10.0 ** rng.uniform(-decades / 2, decades / 2, (1, 1, v.shape[2])) if k.endswith('sd1.weight') else 1.0
            st[k] = (v * rows * cols).astype(np.float32)
        elif k.endswith('sd0_bn.weight'):
            st[k] = (rng.uniform(0.5, gain_max, v.shape) * rng.choice([-1.0, 1.0], v.shape)).astype(np.float32)
        elif k.endswith('film_w1.bias'):                  # a = eps + exp(.): half the features ~1e-7 + eps .. 3e-4, half O(1)
            tiny = rng.random(v.shape) < 0.5
            st[k] = np.where(tiny, rng.uniform(a_floor, a_floor / 2, v.shape), rng.uniform(-2.0, 0.5, v.shape)).astype(np.float32)
    # calibrate: batch statistics of one train-mode pass become the running statistics (momentum 0.1 -> solve for the batch values)
    new = {}
    fo.decoder_forward(p.astype(np.float64), g.astype(np.float64), state64(st), L, 'direct', training=True, new_stats=new)
    for k, v in new.items():
        if k.endswith('running_mean') or k.endswith('running_var'):
            st[k] = ((v - 0.9 * st[k].astype(np.float64)) / 0.1).astype(np.float32)
    return st


@pytest.mark.parametrize('f', [19, 37, 64])
@pytest.mark.parametrize('xscale', [0.3, 50.0])
def test_adversarial_ranges_eval(f, xscale):
    L, G, B, N = 2, 32, 4, 300
    m, st = decoder_and_state(L, f, G, 4000 + f)
    p, g = synth_inputs(B, N, G, 4100 + f)
    p = (p / 0.3 * xscale / 3.0).astype(np.float32).clip(-xscale, xscale)        # |x| up to xscale
    st = adversarial_state(st, p, g, L, 4200 + f)
    rv = np.concatenate([v.ravel() for k, v in st.items() if k.endswith('running_var')])
    assert rv.min() < 1e-4 and rv.max() > 1e3                                     # the variances really span the range
    m = load(m, st).to(DEV).eval()
    for mode in ('direct', 'inverse'):
        o64, l64 = fo.decoder_fused(p.astype(np.float64), g.astype(np.float64), state64(st), L, mode)
        o32, l32 = fo.decoder_fused(p, g, st, L, mode)
        assert np.isfinite(o64).all()
        with torch.no_grad():
            out, ld = m.forward_fused(dev(p), dev(g), mode)
            ps, mus, lvs = m(dev(p), dev(g), mode)
        tol_c, tol_l = tol_at_depth(3 * L, max(np.abs(o64).max(), np.abs(p).max()))
        errs = dict(hip_coord=maxabs(host(out), o64), hip_logdet=maxabs(host(ld), l64), oracle32_coord=maxabs(o32, o64),
                    oracle32_logdet=maxabs(l32, l64), tol_coord=tol_c, tol_logdet=tol_l, xmax=np.abs(o64).max())
        record_parity(f'gpu:adversarial_eval:f{f}:x{xscale}:{mode}', **errs)
        assert errs['hip_coord'] < tol_c and errs['hip_logdet'] < tol_l
        assert errs['hip_coord'] < 3 * errs['oracle32_coord'] + TOL_COORD / 4
        assert errs['hip_logdet'] < 3 * errs['oracle32_logdet'] + TOL_LOGDET / 4
        assert maxabs(host(sum(lvs)), l64) < tol_l


def test_adversarial_ranges_gradients():
    """Same weights through the differentiable eval path (HIP forward + HIP backward): gradients against CPU autograd of
    the torch port, relative to each tensor's gradient norm."""
    from oracle import torch_port as tp
    L, f, G, B, N = 1, 37, 32, 3, 200
    m, st = decoder_and_state(L, f, G, 4300)
    p, g = synth_inputs(B, N, G, 4301)
    st = adversarial_state(st, p, g, L, 4302, gain_max=10.0, decades=2.0, a_floor=-8.0)
    m = load(m, st).to(DEV).eval()
    pd, gd = dev(p).requires_grad_(True), dev(g).requires_grad_(True)
    out, ld = m.forward_fused(pd, gd, 'inverse')
    (0.5 * (out ** 2).sum() / B + 0.5 * ld.sum() / B).backward()
    tst = {k: torch.from_numpy(v).double().requires_grad_(v.dtype == np.float32 and 'running' not in k and not k.endswith('eps'))
           for k, v in st.items()}
    pt, gt = torch.from_numpy(p).double().requires_grad_(True), torch.from_numpy(g).double().requires_grad_(True)
    o_ref, l_ref = tp.decoder_fused(pt, gt, tst, L, 'inverse', grad=True)
    (0.5 * (o_ref ** 2).sum() / B + 0.5 * l_ref.sum() / B).backward()
    worst = 0.0
    named = dict(m.named_parameters())
    for k, v in [('p', pd), ('g', gd)] + list(named.items()):
        ref = (pt if k == 'p' else gt if k == 'g' else tst[k]).grad
        if ref is None:
            continue
        rel = float((v.grad.double().cpu() - ref).norm() / (ref.norm() + 1e-30))
        worst = max(worst, rel)
        assert rel < 2e-4, (k, rel)
    record_parity('gpu:adversarial_gradients:f37', worst_rel=worst)


def _small(f=19):
    L, G, B, N = 2, 16, 3, 130
    m, st = decoder_and_state(L, f, G, 4400)
    p, g = synth_inputs(B, N, G, 4401)
    return L, f, G, B, N, m, st, p, g


@pytest.mark.parametrize('mode', ['direct', 'inverse'])
@pytest.mark.parametrize('bad', [np.nan, np.inf, -np.inf, 1e5])
@pytest.mark.parametrize('f', [19, 37])          # 37: the abs-form widths (|.| source modifiers instead of v_max)
def test_nonfinite_or_out_of_range_points_reach_out_and_logdet(mode, bad, f, monkeypatch):
    """The split-f16 stack kernel on its own (GWTF_NO_RANGE_RERUN=1, what rounds 1-4 shipped): NaN / Inf points and points beyond
    the operand range are flagged NaN.  With the re-run launch (the default) the out-of-range ones come back finite:
    tests/test_gpu_exact.py::test_out_of_range_points_come_back_finite_and_equal_to_the_fp64_oracle."""
    monkeypatch.setenv('GWTF_NO_RANGE_RERUN', '1')
    L, f, G, B, N, m, st, p, g = _small(f)
    m = m.to(DEV).eval()
    p = p.copy()
    hits = [(0, 0, 5), (1, 2, 77), (2, 1, 129)]
    for b, d, n in hits:
        p[b, d, n] = bad
    with torch.no_grad():
        out, ld = m.forward_fused(dev(p), dev(g), mode)
        ps, mus, lvs = m(dev(p), dev(g), mode)
    o, l = host(out), host(ld)
    mask = np.zeros((B, N), bool)
    for b, d, n in hits:
        mask[b, n] = True
        assert not np.isfinite(o[b, :, n]).any() and not np.isfinite(l[b, :, n]).any(), (b, n, o[b, :, n], l[b, :, n])
    # every other point is untouched: equal to the run without the bad points
    p_ok = np.where(np.isfinite(p) & (np.abs(p) < 1e4), p, 0.0).astype(np.float32)
    with torch.no_grad():
        out_ok, ld_ok = m.forward_fused(dev(p_ok), dev(g), mode)
    assert np.array_equal(o[:, :, ][np.broadcast_to(~mask[:, None, :], o.shape)], host(out_ok)[np.broadcast_to(~mask[:, None, :], o.shape)])
    assert np.isfinite(l[np.broadcast_to(~mask[:, None, :], l.shape)]).all()
    # the list API (what the reference's loss reads: ps[0] / ps[-1] and sum(logvars)) carries the flag too
    fin = ps[0] if mode == 'inverse' else ps[-1]
    assert torch.equal(torch.isnan(fin), torch.isnan(out)) and not np.isfinite(host(fin)[0, :, 5]).any()
    assert not np.isfinite(host(sum(lvs))[0, :, 5]).any()


@pytest.mark.parametrize('bad', [np.nan, np.inf])
@pytest.mark.parametrize('f', [19, 37])          # 37: the abs-form widths (|.| source modifiers instead of v_max)
def test_nonfinite_latent_reaches_every_point_of_its_shape(bad, f):
    L, f, G, B, N, m, st, p, g = _small(f)
    m = m.to(DEV).eval()
    g = g.copy()
    g[1, 3] = bad
    with torch.no_grad():
        out, ld = m.forward_fused(dev(p), dev(g), 'inverse')
    o, l = host(out), host(ld)
    assert not np.isfinite(o[1]).any() and not np.isfinite(l[1]).any()
    assert np.isfinite(o[[0, 2]]).all() and np.isfinite(l[[0, 2]]).all()


@pytest.mark.parametrize('key', ['flows.0.nvp2.T_mu_0.mu_sd0.weight', 'flows.1.nvp1.T_logvar_0.logvar_sd1.weight',
                                 'flows.0.nvp3.T_mu_0_cond_b.mu_sd1_film_b1.bias', 'flows.1.nvp3.T_logvar_0.logvar_sd0_bn.running_var',
                                 'flows.0.nvp1.T_logvar_0_cond_w.logvar_sd1_film_w0.weight', 'flows.1.nvp2.T_mu_1.mu_sd2.weight'])
@pytest.mark.parametrize('f', [19, 37])          # 37: the abs-form widths (|.| source modifiers instead of v_max)
def test_nonfinite_weight_reaches_every_output(key, f):
    """A single NaN anywhere in a coupling's parameters / buffers (diverged training): every point's result is NaN -- also
    for the weights whose NaN the v_max ReLU alone would turn into zeros (sd0, sd1, the FiLM shift head)."""
    L, f, G, B, N, m, st, p, g = _small(f)
    st = {k: v.copy() for k, v in st.items()}
    st[key].reshape(-1)[1] = np.nan
    m = load(m, st).to(DEV).eval()
    with torch.no_grad():
        out, ld = m.forward_fused(dev(p), dev(g), 'inverse')
    assert not np.isfinite(host(out)).any() and not np.isfinite(host(ld)).any()
    # differentiable eval path and train mode: the loss is non-finite as well
    out, ld = m.forward_fused(dev(p).requires_grad_(True), dev(g), 'inverse')
    assert not np.isfinite(float((out ** 2).sum() + ld.sum()))
    m.train()
    with torch.no_grad():
        out, ld = m.forward_fused(dev(p), dev(g), 'inverse')
    assert not np.isfinite(float((out ** 2).sum() + ld.sum()))


@pytest.mark.parametrize('f', [19, 37])          # 37: the abs-form widths (|.| source modifiers instead of v_max)
def test_nonfinite_point_in_train_mode_reaches_the_loss(f):
    L, f, G, B, N, m, st, p, g = _small(f)
    m = m.to(DEV).train()
    p = p.copy()
    p[1, 0, 3] = np.nan
    with torch.no_grad():
        out, ld = m.forward_fused(dev(p), dev(g), 'inverse')
    assert not np.isfinite(float((out ** 2).sum() + ld.sum()))


@pytest.mark.parametrize('mode', ['direct', 'inverse'])
@pytest.mark.parametrize('bad', [np.nan, np.inf, -np.inf])
@pytest.mark.parametrize('f', [19, 37])
def test_nonfinite_points_stay_nan_through_the_rerun_launch(mode, bad, f):
    """Default path (split launch + exact-fp32 re-run of flagged tiles): a genuinely non-finite point is recomputed and is NaN again in
    coordinates AND log-det; the other points of its tile are finite and within tolerance of a run without the bad points."""
    L, f, G, B, N, m, st, p, g = _small(f)
    m = m.to(DEV).eval()
    p = p.copy()
    hits = [(0, 0, 5), (1, 2, 77), (2, 1, 129)]
    for b, d, n in hits:
        p[b, d, n] = bad
    with torch.no_grad():
        out, ld = m.forward_fused(dev(p), dev(g), mode)
        ps, mus, lvs = m(dev(p), dev(g), mode)
    o, l = host(out), host(ld)
    mask = np.zeros((B, N), bool)
    for b, d, n in hits:
        mask[b, n] = True
        assert not np.isfinite(o[b, :, n]).any() and not np.isfinite(l[b, :, n]).any()
    p_ok = np.where(np.isfinite(p), p, 0.0).astype(np.float32)
    with torch.no_grad():
        out_ok, ld_ok = m.forward_fused(dev(p_ok), dev(g), mode)
    sel = np.broadcast_to(~mask[:, None, :], o.shape)
    assert np.isfinite(o[sel]).all() and np.isfinite(l[sel]).all()
    assert maxabs(o[sel], host(out_ok)[sel]) < TOL_COORD and maxabs(l[sel], host(ld_ok)[sel]) < TOL_LOGDET
    fin = ps[0] if mode == 'inverse' else ps[-1]
    assert torch.equal(torch.isnan(fin), torch.isnan(out)) and not np.isfinite(host(sum(lvs))[0, :, 5]).any()
