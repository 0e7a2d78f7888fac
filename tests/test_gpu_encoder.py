"""Fused PointNet encoder kernel (csrc/gwtf_encoder.hip through the C ABI) against the reference's golden outputs and the
fp64 oracle.  Needs an MI355X.  Bar: |delta| <= 2e-5 x max(1, |features|_max) -- the same fp32 noise level the reference
shows against fp64 (the split-f16 contraction carries 22 mantissa bits per operand, fp32 accumulation)."""
import numpy as np
import pytest
import torch

from conftest import golden
from helpers import maxabs
from go_with_the_flows_amd import encoders
from go_with_the_flows_amd.synth import load_synth_, synth_inputs
from oracle import encoder_oracle as eo

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
CASES = (('big', [128, 256, 512], 1100), ('small', [128, 64, 128], 1110))


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def host(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize('tag,n_features,seed', CASES)
def test_eval_matches_reference_golden(tag, n_features, seed):
    G = golden('g11_encoder')
    m = encoders.PointNetCloudEncoder(3, 64, n_features)
    load_synth_(m, seed)
    m = m.to(DEV).eval()
    x = dev(G[f'{tag}_x'])
    with torch.no_grad():
        feat, pooled = m(x), m.forward_max(x)
    scale = max(1.0, float(np.abs(G[f'{tag}_eval_pooled']).max()))
    assert maxabs(host(pooled), G[f'{tag}_eval_pooled']) < 2e-5 * scale
    assert maxabs(host(feat[:, :, :6]), G[f'{tag}_eval_feat_head']) < 2e-5 * scale
    assert torch.equal(feat.max(2)[0], pooled)                       # the pooled path is the same arithmetic


@pytest.mark.parametrize('tag,n_features,seed', CASES)
def test_train_mode_and_gradient_path_match_reference_golden(tag, n_features, seed):
    """Batch-statistic BatchNorm runs as library GEMM + batch-norm on the device; running statistics update."""
    G = golden('g11_encoder')
    m = encoders.PointNetCloudEncoder(3, 64, n_features)
    load_synth_(m, seed)
    m = m.to(DEV).train()
    x = dev(G[f'{tag}_x'])
    pooled = m.forward_max(x)
    scale = max(1.0, float(np.abs(G[f'{tag}_train_pooled']).max()))
    assert maxabs(host(pooled), G[f'{tag}_train_pooled']) < 5e-5 * scale
    sd = m.state_dict()
    last = f'features.sd{len(n_features) - 1}_bn.'
    assert maxabs(host(sd[last + 'running_mean']), G[f'{tag}_rm_last']) < 1e-4
    assert maxabs(host(sd[last + 'running_var']), G[f'{tag}_rv_last']) < 1e-4 * max(1, float(G[f'{tag}_rv_last'].max()))
    pooled.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    # eval + requires_grad input: differentiable path, same values as the fused kernel
    m.eval()
    xg = x.clone().requires_grad_()
    y = m.forward_max(xg)
    with torch.no_grad():
        z = m.forward_max(x)
    assert y.requires_grad and maxabs(host(y), host(z)) < 2e-5 * scale


@pytest.mark.parametrize('B,N', [(1, 1), (2, 31), (3, 257), (1, 700), (2, 2048)])
def test_ragged_sizes_against_oracle(B, N):
    m = encoders.PointNetCloudEncoder(3, 64, [128, 256, 512])
    st = load_synth_(m, 77)
    m = m.to(DEV).eval()
    x, _ = synth_inputs(B, N, 4, 78)
    with torch.no_grad():
        feat, pooled = m(dev(x)), m.forward_max(dev(x))
    ref = eo.pointnet_features(x, st, 3)
    scale = max(1.0, float(np.abs(ref).max()))
    assert maxabs(host(feat), ref) < 2e-5 * scale
    assert maxabs(host(pooled), ref.max(2)) < 2e-5 * scale


def test_packed_weights_follow_parameter_updates():
    m = encoders.PointNetCloudEncoder(3, 64, [128, 256, 512])
    load_synth_(m, 5)
    m = m.to(DEV).eval()
    x = dev(synth_inputs(2, 100, 4, 6)[0])
    with torch.no_grad():
        a = m.forward_max(x)
        m.features.sd2.weight.mul_(1.5)            # in-place edit bumps the version counter
        b = m.forward_max(x)
        m.features.sd2.weight.div_(1.5)
        c = m.forward_max(x)
    assert not torch.allclose(a, b) and torch.allclose(a, c, rtol=1e-5, atol=1e-5)


def test_full_size_properties():
    """B=64 x N=2048 (the airplane batch): permutation invariance of the pooled code and agreement of the pooled path with
    the materialised features."""
    m = encoders.PointNetCloudEncoder(3, 64, [128, 256, 512])
    load_synth_(m, 9)
    m = m.to(DEV).eval()
    x = dev(synth_inputs(64, 2048, 4, 10)[0])
    with torch.no_grad():
        pooled = m.forward_max(x)
        perm = torch.randperm(2048, device=DEV)
        pooled_p = m.forward_max(x[:, :, perm].contiguous())
        feat = m(x)
    assert torch.equal(pooled, pooled_p)            # max is order-independent and each point's arithmetic is identical
    assert torch.equal(feat.max(2)[0], pooled)
    assert float(pooled.min()) >= 0


# ---- train-mode pipeline (csrc/gwtf_encoder_train.hip): batch-statistic BatchNorm forward + max-pool + the whole backward ----
def _train_case(tag):
    G = golden('g17_encoder_train')
    seed = {'a': 1700, 'b': 1710}[tag]
    m = encoders.PointNetCloudEncoder(3, 64, [128, 256, 512])
    load_synth_(m, seed)
    return G, m.to(DEV).train(), dev(G[f'{tag}_x']), dev(G[f'{tag}_wgt'])


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_train_pipeline_matches_reference_golden(tag):
    """Pooled code, running statistics of all four BatchNorms and the gradient of every parameter against the genuine
    reference (fp32 run; its own fp64 run gives the noise floor: pooled 4e-6, gradients 1e-6 relative)."""
    G, m, x, wgt = _train_case(tag)
    assert m._train_pipeline_ok(x)                                   # the HIP pipeline is what runs
    pooled = m.forward_max(x)
    assert pooled.grad_fn is not None and 'EncoderTrainFn' in type(pooled.grad_fn).__name__
    ref64 = G[f'{tag}_pooled_f64']
    scale = max(1.0, float(np.abs(ref64).max()))
    err, ref_err = maxabs(host(pooled), ref64), maxabs(G[f'{tag}_pooled'], ref64)
    assert err < 2e-5 * scale and err < max(3 * ref_err, 1e-5 * scale)
    for name, buf in m.named_buffers():
        want = G[f'{tag}_buf.{name}']
        if name.endswith('num_batches_tracked'):
            assert int(buf) == int(want)
        else:
            assert maxabs(host(buf), want) < 2e-5 * max(1.0, float(np.abs(want).max())), name
    (pooled * wgt).sum().backward()
    worst = {}
    for name, prm in m.named_parameters():
        want = G[f'{tag}_grad_f64.{name}']
        gs = float(np.abs(want).max())
        worst[name] = maxabs(host(prm.grad), want) / gs
        assert worst[name] < 2e-4, (name, worst[name])               # split-f16 operands carry 22 bits; sums over B N points
    from conftest import record_parity
    record_parity(f'encoder_train_{tag}', pooled=err, pooled_ref32=ref_err, grad_rel_max=max(worst.values()))


def test_train_pipeline_against_library_path_at_ragged_and_full_sizes():
    """Same module through the library path (rocBLAS GEMM + MIOpen batch-norm + autograd, an independent implementation):
    N not a multiple of the 256-point tile, and the airplane batch 64 x 2048."""
    for B, N, seed in ((2, 1000, 31), (64, 2048, 32)):
        m = encoders.PointNetCloudEncoder(3, 64, [128, 256, 512])
        load_synth_(m, seed)
        m = m.to(DEV).train()
        x = dev(synth_inputs(B, N, 4, seed + 1)[0])
        wgt = torch.randn(B, 512, device=DEV, generator=torch.Generator(DEV).manual_seed(seed))
        import copy
        lib = copy.deepcopy(m)
        pooled = m.forward_max(x)
        (pooled * wgt).sum().backward()
        ref = torch.max(lib.features(x), dim=2)[0]
        (ref * wgt).sum().backward()
        scale = max(1.0, float(ref.detach().abs().max()))
        assert float((pooled - ref).detach().abs().max()) < 5e-5 * scale
        for (name, p), q in zip(m.named_parameters(), lib.parameters()):
            gs = float(q.grad.abs().max())
            assert float((p.grad - q.grad).abs().max()) < 1e-3 * gs, (B, N, name)
        for (name, a), b in zip(m.named_buffers(), lib.buffers()):
            assert torch.allclose(a.float(), b.float(), rtol=1e-4, atol=1e-5), name


def test_train_pipeline_small_gradients_keep_their_precision():
    """A loss scaled by 1e-7 (gradients near the f16 subnormals): the power-of-two operand scale keeps the relative error."""
    G, m, x, wgt = _train_case('a')
    (m.forward_max(x) * wgt * 1e-7).sum().backward()
    for name, prm in m.named_parameters():
        want = G[f'a_grad_f64.{name}'] * 1e-7
        assert maxabs(host(prm.grad), want) / float(np.abs(want).max()) < 2e-4, name


def test_train_pipeline_falls_back_to_the_library_path_outside_its_cover():
    m = encoders.PointNetCloudEncoder(3, 64, [128, 256, 512]).to(DEV).train()
    x = dev(synth_inputs(2, 70, 4, 3)[0])                            # N % 4 != 0
    assert not m._train_pipeline_ok(x)
    assert m.forward_max(x).shape == (2, 512)
    x4 = dev(synth_inputs(2, 72, 4, 3)[0]).requires_grad_()          # gradient wanted for the points themselves
    assert not m._train_pipeline_ok(x4)
    m.forward_max(x4).sum().backward()
    assert x4.grad is not None
    m.features.sd1_bn.momentum = None                                # cumulative-average BatchNorm
    assert not m._train_pipeline_ok(x4.detach())


def test_two_rank_syncbn_encoder_matches_single_process(tmp_path):
    """SyncBatchNorm semantics of the train pipeline (reference train_ae.py:152): two ranks with half of the batch each, the
    statistic sums all-reduced between a layer's kernel and its fold, == one process with the whole batch."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, GWTF_TMP=str(tmp_path), MASTER_ADDR='127.0.0.1')
    port = 29700 + os.getpid() % 1000
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(os.path.dirname(__file__), 'dist_enc_worker.py')]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'ENC2' in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_train_pipeline_propagates_non_finite_weights():
    """A diverged encoder weight must reach the loss as NaN (the reference's isnan guard, training.py:43-46), not be laundered
    by the max-pool keys or a ReLU compiled without NaN semantics."""
    for bad in (float('nan'), float('inf')):
        m = encoders.PointNetCloudEncoder(3, 64, [128, 256, 512])
        load_synth_(m, 51)
        m = m.to(DEV).train()
        with torch.no_grad():
            m.features.sd1.weight[0, 3, 5] = bad
        pooled = m.forward_max(dev(synth_inputs(2, 64, 4, 52)[0]))
        assert not bool(torch.isfinite(pooled).all())
        # a graphed training step runs the backward unconditionally: it must get through (arg-max indices of all-NaN rows are
        # clamped) and leave the weights' gradients visibly non-finite
        pooled.sum().backward()
        torch.cuda.synchronize()
        assert any(not bool(torch.isfinite(p.grad).all()) for p in m.parameters())


@pytest.mark.parametrize('B,N', [(1, 4), (1, 36), (5, 132), (2, 2500), (3, 516)])
def test_train_pipeline_odd_shapes_against_library_path(B, N):
    """Cloud sizes around every tile boundary of the pipeline (32-point k-steps, 128/256-point workgroups, the SVR config's
    2500 points, a single shape): pooled code, arg-max, gradients and running statistics against the library path."""
    import copy
    m = encoders.PointNetCloudEncoder(3, 64, [128, 256, 512])
    load_synth_(m, 60 + N)
    m = m.to(DEV).train()
    lib = copy.deepcopy(m)
    x = dev(synth_inputs(B, N, 4, 61 + N)[0])
    wgt = torch.randn(B, 512, device=DEV, generator=torch.Generator(DEV).manual_seed(N))
    assert m._train_pipeline_ok(x)
    pooled, amax = m.forward_max(x, return_indices=True)
    (pooled * wgt).sum().backward()
    feat = lib.features(x)
    ref, ref_idx = torch.max(feat, dim=2)
    (ref * wgt).sum().backward()
    scale = max(1.0, float(ref.detach().abs().max()))
    assert float((pooled - ref).detach().abs().max()) < 5e-5 * scale
    # the arg-max may differ only where two points tie to rounding: the feature at OUR index must be the maximum too
    picked = torch.gather(feat.detach(), 2, amax.long().unsqueeze(2)).squeeze(2)
    assert float((picked - ref.detach()).abs().max()) < 5e-5 * scale
    for (name, p), q in zip(m.named_parameters(), lib.parameters()):
        gs = float(q.grad.abs().max())
        assert float((p.grad - q.grad).abs().max()) < 2e-3 * gs + 1e-6, (B, N, name)
    for (name, a), b in zip(m.named_buffers(), lib.buffers()):
        assert torch.allclose(a.float(), b.float(), rtol=1e-4, atol=1e-5), name


def test_train_pipeline_with_negative_batchnorm_weights_pools_the_minimum():
    """BatchNorm + ReLU + max over the points falls with y where the layer's BatchNorm weight is negative: the top layer's forward kernel
    tracks the arg-MIN of y for those channels (one extreme per channel, chosen by the sign of gamma).  Half of the top layer's (and some
    of the lower layers') weights negated; outputs, arg-max gradients and running statistics against the library path."""
    import copy
    B, N, seed = 3, 700, 77
    m = encoders.PointNetCloudEncoder(3, 64, [128, 256, 512])
    load_synth_(m, seed)
    with torch.no_grad():
        for name, prm in m.named_parameters():
            if name.endswith('_bn.weight'):
                prm[::2].neg_()
                if 'sd2' in name:
                    prm[5] = 0.0                                    # a zero weight: the channel is constant, any point is "the" maximum
                    prm[7] = -1e-42                                 # a denormal negative weight: gamma * rstd flushes to zero
    m = m.to(DEV).train()
    x = dev(synth_inputs(B, N, 4, seed + 1)[0])
    wgt = torch.randn(B, 512, device=DEV, generator=torch.Generator(DEV).manual_seed(seed))
    lib = copy.deepcopy(m)
    pooled = m.forward_max(x)
    (pooled * wgt).sum().backward()
    ref = torch.max(lib.features(x), dim=2)[0]
    (ref * wgt).sum().backward()
    scale = max(1.0, float(ref.detach().abs().max()))
    assert float((pooled - ref).detach().abs().max()) < 5e-5 * scale
    for (name, p), q in zip(m.named_parameters(), lib.parameters()):
        gs = float(q.grad.abs().max()) + 1e-12
        assert float((p.grad - q.grad).abs().max()) < 1e-3 * gs, name
    for (name, a), b in zip(m.named_buffers(), lib.buffers()):
        assert torch.allclose(a.float(), b.float(), rtol=1e-4, atol=1e-5), name
