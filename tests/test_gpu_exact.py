"""The exact-fp32 contraction body (csrc/gwtf_stack_exact.hip: v_mfma_f32_16x16x4_f32 on unsplit operands) against the goldens, the
on-device A/B "split-f16 vs exact" on every golden and on bench.py's full grids, and the re-run of out-of-range tiles.  Needs an MI355X.

What the A/B pins (missing #3): the claim that three f16 MFMA products of hi / lo split operands are fp32-grade was
inferred from errors against fp64; here both contraction bodies run on the device on the same records and are compared with each
other and with the reference's own fp64 results -- the measured errors go to gpurun_out/parity_errors.jsonl (copied to profiles/)."""
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, golden, TOL_COORD, TOL_LOGDET, tol_at_depth, record_parity
from helpers import decoder_and_state, maxabs
import go_with_the_flows_amd as gw
from go_with_the_flows_amd import _lib
from go_with_the_flows_amd.synth import load_synth_, synth_inputs
from oracle import torch_port as tp

sys.path.insert(0, ROOT)
import bench                                                   # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
CASES = ['g3_decoder_4x64x128', 'g3s_decoder_lists', 'g4_width37', 'g4_width33', 'g4_width19', 'g16_width80', 'g16_width96',
         'g16_width100', 'g16_width128', 'g15_depth_11x37x128', 'g15_depth_11x33x512', 'g15_depth_6x19x128']


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def host(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize('name', CASES)
@pytest.mark.parametrize('mode', ['direct', 'inverse'])
def test_exact_body_on_every_decoder_golden_and_ab_against_the_split_body(name, mode):
    D = golden(name)
    L, f, G, B, N, seed = D['dims']
    m, _ = decoder_and_state(L, f, G, seed)
    m = m.to(DEV).eval()
    _lib.set_tuning(0)
    tag = f'eval_{mode}'
    pd, gd = dev(D['p']), dev(D['g'])
    with torch.no_grad():
        s_out, s_ld = m.forward_fused(pd, gd, mode=mode)
        s_ps, s_mus, s_lvs = m(pd, gd, mode=mode)
        with _lib.exact_fp32():
            x_out, x_ld = m.forward_fused(pd, gd, mode=mode)
            x_ps, x_mus, x_lvs = m(pd, gd, mode=mode)
    assert torch.equal(x_out, x_ps[0] if mode == 'inverse' else x_ps[-1])                # fused == list variant of the exact body
    ref32 = D[('first_' if mode == 'inverse' else 'last_') + tag]
    ref64 = D[('first64_' if mode == 'inverse' else 'last64_') + tag]
    tol_c, tol_l = tol_at_depth(3 * L, max(np.abs(ref64).max(), np.abs(D['p']).max()))
    e = dict(exact_vs_ref64_coord=maxabs(host(x_out), ref64), split_vs_ref64_coord=maxabs(host(s_out), ref64),
             exact_vs_ref64_logdet=maxabs(host(x_ld), D['logdet64_' + tag]), split_vs_ref64_logdet=maxabs(host(s_ld), D['logdet64_' + tag]),
             ref32_vs_ref64_coord=maxabs(ref32, ref64), ref32_vs_ref64_logdet=maxabs(D['logdet_' + tag], D['logdet64_' + tag]),
             split_vs_exact_coord=maxabs(host(s_out), host(x_out)), split_vs_exact_logdet=maxabs(host(s_ld), host(x_ld)),
             tol_coord=tol_c, tol_logdet=tol_l)
    record_parity(f'exact_ab:{name}:{mode}', **e)
    assert e['exact_vs_ref64_coord'] < tol_c and e['exact_vs_ref64_logdet'] < tol_l
    assert maxabs(host(x_out), ref32) < tol_c and maxabs(host(x_ld), D['logdet_' + tag]) < tol_l
    assert e['split_vs_exact_coord'] < tol_c and e['split_vs_exact_logdet'] < tol_l
    # the split body is fp32-grade: never further from the reference's fp64 run than twice the exact-fp32 body (+ a quarter of the bar)
    assert e['split_vs_ref64_coord'] < 2 * e['exact_vs_ref64_coord'] + tol_c / 4
    assert e['split_vs_ref64_logdet'] < 2 * e['exact_vs_ref64_logdet'] + tol_l / 4
    assert maxabs(host(sum(x_lvs)), D['logdet64_' + tag]) < tol_l
    if 'ps_' + tag in D.files:                                                           # every list slot of the exact body
        assert maxabs(host(torch.stack(x_ps)), D['ps_' + tag]) < TOL_COORD
        assert maxabs(host(torch.stack(x_mus)), D['mus_' + tag]) < TOL_COORD
        assert maxabs(host(torch.stack(x_lvs)), D['lvs_' + tag]) < TOL_LOGDET


@pytest.mark.parametrize('state', ['bench', 'conditioned'])
@pytest.mark.parametrize('name', sorted(bench.WORKLOADS))
def test_split_vs_exact_on_the_bench_grids_every_shape_every_component(name, state):
    """Both bodies on bench.py's exact grids, compared on the device (every shape, every component), per point and relative to the
    coordinate's magnitude as in tests/test_gpu_fullgrid.py.  'conditioned' (output layers x synth.CONDITIONED_GAIN, |z| of a few
    units as a trained model gives): the two bodies differ by less than the stated bar at this depth.  'bench' (the timed state: some
    inverse points reach |x| ~ 5e3, where two fp32 evaluations of the SAME function differ by 2e-4 -- test_gpu_fullgrid's fp32 oracle
    against fp64): recorded, and bounded by a few times that fp32 noise."""
    from go_with_the_flows_amd.synth import CONDITIONED_GAIN
    gain = 1.0 if state == 'bench' else CONDITIONED_GAIN
    cfg = bench.WORKLOADS[name]
    K, L, f, G, B, N, mode = (cfg[k] for k in ('K', 'L', 'f', 'G', 'B', 'N', 'mode'))
    _lib.set_tuning(0)
    decs = []
    for k in range(K):
        d = gw.LocalCondRNVPDecoder(L, f, G)
        load_synth_(d, 2 + k, gain)
        decs.append(d.to(DEV).eval())
    p, g = synth_inputs(B, N, G, 0)
    pd, gd = dev(p), dev(g)
    stack = gw.MixtureStack(decs)
    sideways = mode == 'direct' and K > 1
    counts = [N // K] * K
    run = (lambda: stack.forward_partition(pd, gd, counts, mode)) if sideways else (lambda: stack.forward_all(pd, gd, mode))
    with torch.no_grad():
        sz, sld = run()
        with _lib.exact_fp32():
            xz, xld = run()
            xz2, _ = run()
    assert torch.equal(xz, xz2)                                                          # the exact body repeats bit for bit
    if sideways:
        sz, sld, xz, xld = (t[:, :, :K * (N // K)] for t in (sz, sld, xz, xld))
    assert torch.isfinite(xz).all() and torch.isfinite(xld).all()
    mag = xz.abs().amax(dim=-2, keepdim=True)
    dc = float(((sz - xz).abs() / torch.clamp(mag / 6.0, min=1.0)).max())
    dl = float(((sld - xld).abs() / torch.clamp(mag / 24.0, min=1.0)).max())
    tol_c, tol_l = tol_at_depth(3 * L, 1.0)
    record_parity(f'exact_ab_grid:{state}:{name}:K{K}_f{f}_{B}x{N}_{mode}', split_vs_exact_coord=dc, split_vs_exact_logdet=dl,
                  xmax=float(mag.max()), tol_coord=tol_c, tol_logdet=tol_l)
    if state == 'conditioned':
        assert dc < tol_c and dl < tol_l, (dc, dl, tol_c, tol_l)
    else:
        assert dc < 10 * tol_c and dl < 10 * tol_l, (dc, dl, tol_c, tol_l)


@pytest.mark.parametrize('mode', ['direct', 'inverse'])
@pytest.mark.parametrize('f', [19, 37, 64])
def test_out_of_range_points_come_back_finite_and_equal_to_the_fp64_oracle(mode, f):
    """|x| = 1e5 is beyond the split-f16 body's operand range (GWTF_X_LIMIT = 3e4: the stack kernel flags such a point NaN); the
    re-run launch recomputes the flagged tiles on the exact-fp32 body, so the caller sees what the reference computes (flows.py:
    113-115 has no range limit): finite values equal to the fp64 oracle within fp32 rounding of a coordinate of that size."""
    L, G, B, N = 2, 16, 3, 300
    m, st = decoder_and_state(L, f, G, 4400)
    m = m.to(DEV).eval()
    p, g = synth_inputs(B, N, G, 4401)
    hits = [(0, 0, 5), (1, 2, 77), (2, 1, 299)]
    for i, (b, d, n) in enumerate(hits):
        p[b, d, n] = (1e5, -4e4, 7.5e5)[i]
    with torch.no_grad():
        out, ld = m.forward_fused(dev(p), dev(g), mode)
        ps, mus, lvs = m(dev(p), dev(g), mode)
    tst = {k: (torch.from_numpy(v).double() if v.dtype == np.float32 else torch.from_numpy(v)) for k, v in st.items()}
    z64, ld64 = tp.decoder_fused(torch.from_numpy(p).double(), torch.from_numpy(g).double(), tst, L, mode)
    z64, ld64 = z64.numpy(), ld64.numpy()
    o, l = host(out), host(ld)
    assert np.isfinite(o).all() and np.isfinite(l).all()
    mag = np.maximum(np.abs(z64).max(axis=1, keepdims=True), np.abs(p).max(axis=1, keepdims=True))
    tol_c, tol_l = tol_at_depth(3 * L, 1.0)
    ec = float((np.abs(o - z64) / np.maximum(1.0, mag / 6.0)).max())
    el = float((np.abs(l - ld64) / np.maximum(1.0, mag / 24.0)).max())
    record_parity(f'range_rerun:f{f}:{mode}', coord=ec, logdet=el, tol_coord=tol_c, tol_logdet=tol_l, xmax=float(mag.max()))
    assert ec < tol_c and el < tol_l, (ec, el)
    fin = ps[0] if mode == 'inverse' else ps[-1]
    assert torch.equal(fin, out) and maxabs(host(sum(lvs)), l) <= 1e-6 * max(1.0, np.abs(l).max())
    assert all(torch.isfinite(t).all() for t in ps + mus + lvs)                        # every list slot of the re-run tiles
    # tiles without an out-of-range point are not touched by the re-run: bit-equal to a launch without it
    import os
    os.environ['GWTF_NO_RANGE_RERUN'] = '1'
    try:
        with torch.no_grad():
            o_off, l_off = (host(t) for t in m.forward_fused(dev(p), dev(g), mode))
    finally:
        del os.environ['GWTF_NO_RANGE_RERUN']
    for b, d, n in hits:
        assert not np.isfinite(o_off[b, :, n]).any() and not np.isfinite(l_off[b, :, n]).any()    # what rounds 1-4 returned
    clean = np.ones((B, N), bool)
    for b, d, n in hits:
        clean[b, (n // 128) * 128:(n // 128 + 1) * 128] = False                         # the largest tile a hit can sit in (f <= 64: 64..256 points)
        clean[b, (n // 256) * 256:(n // 256 + 1) * 256] = False
    sel = np.broadcast_to(clean[:, None, :], o.shape)
    assert np.array_equal(o[sel], o_off[sel]) and np.array_equal(l[sel], l_off[sel])


def test_exact_body_of_a_mixture_partition_and_direct_mode():
    """K components on a partition of the points (the sampling path) through the exact body == each component's own exact launch."""
    K, L, f, G, B, N = 3, 2, 19, 16, 2, 500
    decs = []
    for k in range(K):
        d = gw.LocalCondRNVPDecoder(L, f, G)
        load_synth_(d, 60 + k)
        decs.append(d.to(DEV).eval())
    p, g = synth_inputs(B, N, G, 9)
    pd, gd = dev(p), dev(g)
    counts = [200, 0, 300]
    stack = gw.MixtureStack(decs)
    with torch.no_grad(), _lib.exact_fp32():
        x, ld = stack.forward_partition(pd, gd, counts, 'direct')
        off = 0
        for k, c in enumerate(counts):
            if c:
                xo, lo = decs[k].forward_fused(pd[:, :, off:off + c].contiguous(), gd, 'direct')
                assert torch.equal(x[:, :, off:off + c], xo) and torch.equal(ld[:, :, off:off + c], lo)
            off += c
        split = stack.forward_partition
    with torch.no_grad():
        xs, _ = split(pd, gd, counts, 'direct')
    assert maxabs(host(xs), host(x)) < TOL_COORD


def _worklist(px, K, C, f):
    n = K * C * _lib.lib().gwtf_packed_x_coupling_floats(f)
    assert px.numel() == n + _lib.WORKLIST_INTS
    return px[n:].view(torch.int32)


@pytest.mark.parametrize('hits', ['few', 'more_than_the_list_holds'])
def test_work_list_of_the_rerun_pair(hits, monkeypatch):
    """The flagging launch lists the waves that flagged a point, the re-run launch walks the list (every tile when the list overflows)
    and leaves it cleared: same results as the re-run that scans every tile's flags (GWTF_NO_RERUN_WORKLIST=1), on a mixture launch
    (K components share the clouds) and on a single decoder, call after call."""
    K, L, f, G, B, N = 2, 1, 19, 16, 40, 2048                    # 2 x 40 x 2048 / 64 = 2560 waves > GWTF_WORKLIST_CAP
    decs = []
    for k in range(K):
        d = gw.LocalCondRNVPDecoder(L, f, G)
        load_synth_(d, 70 + k)
        decs.append(d.to(DEV).eval())
    p, g = synth_inputs(B, N, G, 11)
    if hits == 'few':
        p[0, 0, 3], p[7, 2, 1000], p[39, 1, 2047] = 1e5, -2e5, 4e4
    else:
        p[:, 0, :] = 1e5 * np.sign(p[:, 0, :] + 1e-9)             # every wave of every component flags
    pd, gd = dev(p), dev(g)
    stack = gw.MixtureStack(decs)
    res = []
    with torch.no_grad():
        for rep in range(3):
            z, ld = stack.forward_all(pd, gd, 'inverse')
            z1, ld1 = decs[1].forward_fused(pd, gd, 'inverse')
            assert torch.isfinite(z).all() and torch.isfinite(ld).all()
            assert torch.equal(z[1], z1) and torch.equal(ld[1], ld1)
            res.append((z.clone(), ld.clone()))
            wl = _worklist(stack.packed_exact(), K, 3 * L, f)
            assert int(wl[0]) == 0 and int(wl[1]) == 0, wl[:2]                         # cleared by the re-run launch's last workgroup
            assert int(_worklist(decs[1].engine().packed_exact(), 1, 3 * L, f)[0]) == 0
        assert all(torch.equal(res[0][0], r[0]) and torch.equal(res[0][1], r[1]) for r in res[1:])
        monkeypatch.setenv('GWTF_NO_RERUN_WORKLIST', '1')
        z_scan, ld_scan = stack.forward_all(pd, gd, 'inverse')
        assert torch.equal(z_scan, res[0][0]) and torch.equal(ld_scan, res[0][1])
        with _lib.exact_fp32():                                                          # and the flagged points hold the exact body's values
            z_x, ld_x = stack.forward_all(pd, gd, 'inverse')
    bad = torch.from_numpy(np.abs(p).max(axis=1) > 3e4).to(DEV)                          # (B, N)
    sel = bad[None, :, None, :].expand_as(z_x)
    assert torch.equal(res[0][0][sel], z_x[sel]) and torch.equal(res[0][1][sel], ld_x[sel])
