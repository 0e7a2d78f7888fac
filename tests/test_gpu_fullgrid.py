"""Every bench.py workload at its exact grid -- same B, N, K, seeds and default tile as the timed launch -- with EVERY shape and
EVERY component compared to the fp64 oracle (oracle/torch_port.py).  Needs an MI355X.

Why all shapes: the stack kernel places VALU instructions by hand beside VGPR-form MFMAs (inline asm, outside LLVM's hazard
recogniser); the three hazards met so far (docs/LOG.md 4.1) were shape- and occupancy-dependent -- wrong values in the
last point block of some waves, only with two workgroups resident per compute unit, only for some shapes.  A check that samples
three shapes of 64 can miss that; this one cannot.  Each launch also runs twice and must repeat bit for bit (a timing-dependent
hazard shows up as run-to-run differences before it shows up against the oracle).
"""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, TOL_COORD, TOL_LOGDET, tol_at_depth, record_parity
import go_with_the_flows_amd as gw
from go_with_the_flows_amd import _lib
from go_with_the_flows_amd.synth import load_synth_, synth_inputs
from oracle import torch_port as tp

sys.path.insert(0, ROOT)
import bench                                                   # noqa: E402  (WORKLOADS: the grids the bench line is quoted on)

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _cpu_threads():
    torch.set_num_threads(bench.host_cores())


def _state(st, dtype):
    return {k: (torch.from_numpy(v).to(dtype) if v.dtype == np.float32 else torch.from_numpy(v)) for k, v in st.items()}


def _oracle(p, g, st, L, mode, dtype, training=False):
    z, ld = tp.decoder_fused(torch.from_numpy(p).to(dtype), torch.from_numpy(g).to(dtype), _state(st, dtype), L, mode,
                             training=training)
    return z.numpy(), ld.numpy()


def _bounds(tag, hip_z, hip_ld, z64, ld64, z32, ld32, C, extra=None, absolute=False):
    """Point-wise bar.  The stated tolerance grows with a coordinate's magnitude (conftest.tol_at_depth: rounding error is relative);
    bench.py's synthetic weights send some points of the inverse pass to |x| ~ 5e3, so it is applied PER POINT: error divided by
    max(1, |x|/6) (coordinates) / max(1, |x|/24) (log-det) of that point, against tol(C) = 2e-5 / 1e-5 * max(1, C/12).  Two
    assertions: no further from fp64 than 3x the fp32 evaluation of the same function (the reference's own arithmetic), and inside
    the stated bar wherever the fp32 evaluation itself is."""
    mag = np.abs(z64).max(axis=-2, keepdims=True)
    sc, sl = np.maximum(1.0, mag / 6.0), np.maximum(1.0, mag / 24.0)
    e = dict(hip_vs_fp64_coord=float((np.abs(hip_z - z64) / sc).max()), hip_vs_fp64_logdet=float((np.abs(hip_ld - ld64) / sl).max()),
             fp32_vs_fp64_coord=float((np.abs(z32 - z64) / sc).max()), fp32_vs_fp64_logdet=float((np.abs(ld32 - ld64) / sl).max()),
             hip_vs_fp64_coord_abs=float(np.abs(hip_z - z64).max()), fp32_vs_fp64_coord_abs=float(np.abs(z32 - z64).max()),
             xmax=float(mag.max()), **(extra or {}))
    e['tol_coord'], e['tol_logdet'] = tol_at_depth(C, 1.0)
    if not absolute:
        record_parity('fullgrid:' + tag, **e)
    assert np.isfinite(hip_z).all() and np.isfinite(hip_ld).all(), tag
    if absolute:
        # the well-conditioned state: the STATED tolerance at this depth and coordinate range, absolute, no reference to fp32 noise
        ta, tl = tol_at_depth(C, e['xmax'])
        e['abs_tol_coord'], e['abs_tol_logdet'] = ta, tl
        e['hip_vs_fp64_logdet_abs'] = float(np.abs(hip_ld - ld64).max())
        record_parity('fullgrid-conditioned:' + tag, **e)
        assert e['hip_vs_fp64_coord_abs'] < ta and e['hip_vs_fp64_logdet_abs'] < tl, (tag, e)
        return e
    assert e['hip_vs_fp64_coord'] < 3 * e['fp32_vs_fp64_coord'] + TOL_COORD / 4, (tag, e)
    assert e['hip_vs_fp64_logdet'] < 3 * e['fp32_vs_fp64_logdet'] + TOL_LOGDET / 4, (tag, e)
    assert e['hip_vs_fp64_coord'] < max(e['tol_coord'], 1.5 * e['fp32_vs_fp64_coord']), (tag, e)
    assert e['hip_vs_fp64_logdet'] < max(e['tol_logdet'], 1.5 * e['fp32_vs_fp64_logdet']), (tag, e)
    return e


@pytest.mark.parametrize('state', ['bench', 'conditioned'])
@pytest.mark.parametrize('name', sorted(bench.WORKLOADS))
def test_bench_workload_every_shape_every_component_against_fp64_oracle(name, state):
    """state 'bench': the synthetic weights bench.py times (per-point bar relative to the coordinate's magnitude, never further
    from fp64 than the fp32 evaluation of the same function); 'conditioned': the same draw with the couplings' output layers scaled
    by synth.CONDITIONED_GAIN, which keeps |z| at a few units as a trained model does -- there the stated ABSOLUTE tolerance
    (conftest.tol_at_depth) must hold on every point of the full grid."""
    from go_with_the_flows_amd.synth import CONDITIONED_GAIN
    gain = 1.0 if state == 'bench' else CONDITIONED_GAIN
    cfg = bench.WORKLOADS[name]
    K, L, f, G, B, N, mode = (cfg[k] for k in ('K', 'L', 'f', 'G', 'B', 'N', 'mode'))
    _cpu_threads()
    _lib.set_tuning(0)                 # the library's own tile choice, as in the timed run
    decs, states = [], []
    for k in range(K):                                           # bench.run_workload's construction, rank 0
        d = gw.LocalCondRNVPDecoder(L, f, G)
        states.append(load_synth_(d, 2 + k, gain))
        decs.append(d.to(DEV).eval())
    p, g = synth_inputs(B, N, G, 0)
    pd, gd = torch.from_numpy(p).to(DEV), torch.from_numpy(g).to(DEV)
    stack = gw.MixtureStack(decs)
    sideways = mode == 'direct' and K > 1
    counts = [N // K] * K
    runs = []
    with torch.no_grad():
        for _ in range(2):
            out = stack.forward_partition(pd, gd, counts, mode) if sideways else stack.forward_all(pd, gd, mode)
            runs.append([t.clone() for t in out])
    torch.cuda.synchronize()
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1]), 'two launches of the same grid differ'
    hz, hld = runs[0][0].cpu().numpy(), runs[0][1].cpu().numpy()
    C = 3 * L
    if sideways:                                                 # each point through ONE component: (B,3,N), segment k -> component k
        z64, ld64, z32, ld32 = (np.zeros((B, 3, N), dt) for dt in (np.float64, np.float64, np.float32, np.float32))
        for k in range(K):
            a, b = k * (N // K), (k + 1) * (N // K)
            z64[:, :, a:b], ld64[:, :, a:b] = _oracle(p[:, :, a:b], g, states[k], L, mode, torch.float64)
            z32[:, :, a:b], ld32[:, :, a:b] = _oracle(p[:, :, a:b], g, states[k], L, mode, torch.float32)
        covered = K * (N // K)
        hz, hld, z64, ld64, z32, ld32 = (t[:, :, :covered] for t in (hz, hld, z64, ld64, z32, ld32))
    else:                                                        # every component on every point: (K,B,3,N)
        z64, ld64 = (np.stack(t) for t in zip(*[_oracle(p, g, states[k], L, mode, torch.float64) for k in range(K)]))
        z32, ld32 = (np.stack(t) for t in zip(*[_oracle(p, g, states[k], L, mode, torch.float32) for k in range(K)]))
        assert hz.shape == (K, B, 3, N)
    _bounds(f'{name}:K{K}_f{f}_{B}x{N}_{mode}', hz, hld, z64, ld64, z32, ld32, C, absolute=state == 'conditioned')


@pytest.mark.parametrize('state', ['bench', 'conditioned'])
def test_train_mode_forward_at_airplane_grid_every_shape_every_component(state):
    """model.train() forward (batch-statistic BatchNorm over all 64 x 2048 points; reference flows.py:27,30,62,65) of the airplane
    config's K = 4 decoders through the K-batched pipeline, all shapes and components against the fp64 oracle in train mode, and the
    updated running statistics."""
    from go_with_the_flows_amd.synth import CONDITIONED_GAIN
    K, L, f, G, B, N = 4, 11, 37, 128, 64, 2048
    _cpu_threads()
    _lib.set_tuning(0)
    decs, states = [], []
    for k in range(K):
        d = gw.LocalCondRNVPDecoder(L, f, G)
        states.append(load_synth_(d, 2 + k, 1.0 if state == 'bench' else CONDITIONED_GAIN))
        decs.append(d.to(DEV).train())
    p, g = synth_inputs(B, N, G, 0)
    stack = gw.MixtureStack(decs)
    with torch.no_grad():
        z, ld = stack.forward_all(torch.from_numpy(p).to(DEV), torch.from_numpy(g).to(DEV), 'inverse')
    hz, hld = z.cpu().numpy(), ld.cpu().numpy()
    z64, ld64, z32, ld32, rv_err = [], [], [], [], 0.0
    for k in range(K):
        s64, s32 = _state(states[k], torch.float64), _state(states[k], torch.float32)
        a, b = tp.decoder_fused(torch.from_numpy(p).double(), torch.from_numpy(g).double(), s64, L, 'inverse', training=True)
        c, d_ = tp.decoder_fused(torch.from_numpy(p), torch.from_numpy(g), s32, L, 'inverse', training=True)
        z64.append(a.numpy()); ld64.append(b.numpy()); z32.append(c.numpy()); ld32.append(d_.numpy())
        sd = decs[k].state_dict()
        for key, ref in s64.items():                             # the oracle updated its running statistics in place
            if 'running_' in key:
                rv_err = max(rv_err, float((sd[key].cpu().double() - ref).abs().max() / (ref.abs().max() + 1e-12)))
    z64, ld64, z32, ld32 = (np.stack(t) for t in (z64, ld64, z32, ld32))
    _bounds(f'train_forward:K{K}_f{f}_{B}x{N}_inverse', hz, hld, z64, ld64, z32, ld32, 3 * L, extra={'running_stats_rel': rv_err},
            absolute=state == 'conditioned')
    assert rv_err < 1e-4, rv_err

