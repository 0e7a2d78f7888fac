"""CPU-side checks of the boundary: checkpoint contract, C ABI exports, host logic, loud failure without a GPU."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT
import go_with_the_flows_amd as gw
from go_with_the_flows_amd import _lib
from go_with_the_flows_amd.synth import load_synth_


def test_state_dict_contract_matches_reference():
    spec = json.load(open(os.path.join(GOLDEN, 'contract.json')))
    m = gw.LocalCondRNVPDecoder(*spec['ctor'])
    sd = m.state_dict()
    assert [k for k, _, _ in spec['state_dict']] == list(sd.keys())          # names AND order
    for k, shape, dtype in spec['state_dict']:
        assert list(sd[k].shape) == shape and str(sd[k].dtype) == 'torch.' + dtype, k
    # strict round trip
    m2 = gw.LocalCondRNVPDecoder(*spec['ctor'])
    load_synth_(m, 1)
    m2.load_state_dict(m.state_dict(), strict=True)
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))


def test_constructor_surface():
    d = gw.LocalCondRNVPDecoder(3, 19, 32, weight_std=0.02)
    assert (d.n_flows, d.f_n_features, d.g_n_features, d.weight_std) == (3, 19, 32, 0.02)
    assert len(d.flows) == 3 and [t.pattern for t in d.flows] == [0, 1, 0]
    assert [c.warp_inds for c in d.flows[1].couplings()] == [[0, 1], [0, 2], [1, 2]]
    assert d.flows[0].nvp2.keep_inds == [0, 2]
    assert gw.LocalCondRNVPDecoder.get_param_count(4, 64, 128) == 701952          # reference formula
    assert sum(p.numel() for p in gw.LocalCondRNVPDecoder(4, 64, 128).parameters()) == 705060
    with pytest.raises(ValueError):
        gw.CondRealNVPFlow3D(8, 8, warp_inds=[1, 0])
    with pytest.raises(ValueError):
        gw.CondRealNVPFlow3DTriple(8, 8, pattern=2)
    # near-identity init of the last layers (reference flows.py:52-58,87-93)
    c = gw.CondRealNVPFlow3D(64, 128)
    assert float(c.T_mu_1[1].bias.abs().max()) == 0.0 and float(c.T_mu_1[1].weight.std()) < 0.02
    assert float(c.T_logvar_0_cond_w[3].weight.std()) < 0.02


def test_library_loads_and_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, 'include', 'gwtf.h')).read()
    declared = set(re.findall(r'\b(gwtf_[a-z0-9_]+)\s*\(', header))
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), f'{name} declared in include/gwtf.h but not exported'
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    L = _lib.lib()
    assert L.gwtf_abi_version() == _lib.ABI_VERSION
    assert b'bad argument' in L.gwtf_error_string(10001)


def test_buffer_sizes_and_raw_arena_layout():
    L = _lib.lib()
    for f, G in [(8, 16), (19, 128), (33, 512), (37, 128), (64, 128)]:
        FP = L.gwtf_padded_width(f)
        assert FP == (f + 15) // 16 * 16
        per = 2 * (3 * f * f + 2 * f * G + 20 * f + 2)
        assert L.gwtf_raw_coupling_floats(f, G) == per
        c = gw.CondRealNVPFlow3D(f, G, warp_inds=[0, 2])
        assert sum(t.numel() for t in c.raw_tensors()) == per
        assert L.gwtf_film_out_floats(f) == 6 * FP + 4
        assert L.gwtf_packed_w_coupling_floats(f) % 256 == 0
    d = gw.LocalCondRNVPDecoder(2, 8, 16)
    eng = d.engine()
    assert eng.C == 6 and eng.pattern0 == 0
    assert eng.raw_arena().numel() == 6 * L.gwtf_raw_coupling_floats(8, 16)


def test_no_cpu_fallback_and_bad_arguments():
    d = gw.LocalCondRNVPDecoder(1, 8, 16).eval()
    with pytest.raises(gw.GwtfError):
        d(torch.zeros(2, 3, 5), torch.zeros(2, 16))
    with pytest.raises(ValueError):
        d.engine().run(torch.zeros(2, 4, 5), torch.zeros(2, 16), 'inverse', False)
    with pytest.raises(ValueError):
        d.engine().run(torch.zeros(2, 3, 5), torch.zeros(2, 16), 'sideways', False)
    with pytest.raises(NotImplementedError):
        gw.LocalCondRNVPDecoder(1, 129, 16).engine()
    # bad arguments are rejected by the library itself before anything is launched
    L = _lib.lib()
    assert L.gwtf_stack_forward(None, None, None, None, None, None, None, None, 1, 1, 1, 8, 0, 1e-6, 1, 0, None) == 10001
    assert L.gwtf_pack_weights(None, None, None, 1, 8, 16, 0, 0, None) == 10001
    assert L.gwtf_mixture_nll(None, None, None, None, None, None, None, 65, 1, 1, None) == 10001


def test_packed_cache_invalidation_rules():
    d = gw.LocalCondRNVPDecoder(1, 8, 16)
    eng = d.engine()
    k0 = eng._key(False)
    assert eng._key(False) == k0
    with torch.no_grad():
        d.flows[0].nvp1.T_mu_1[1].bias.add_(1.0)          # tracked in-place edit
    k1 = eng._key(False)
    assert k1 != k0
    d.eval()                                               # train()/eval() bump the stamp (reference optimiser uses .data)
    k2 = eng._key(False)
    assert k2 != k1
    d.load_state_dict(d.state_dict())
    assert eng._key(False) != k2
    d.flows[0].nvp1.invalidate_packed_weights()
    assert eng._key(False) != k2


def test_library_keeps_no_tuning_state_and_the_tile_plan_follows_rounds():
    """the tuning word is an ARGUMENT (no setter is exported any more), and the forward's tile is chosen by
    resident rounds x the calibrated cost of a round (csrc/gwtf_stack.hip tile_cost), not by the point count alone.  Host-only:
    gwtf_stack_plan launches nothing."""
    assert not hasattr(ctypes.CDLL(_lib.LIB_PATH), 'gwtf_debug_set_points_per_wave')
    plan = _lib.stack_plan
    # the bench workloads (K, B, N, f): airplane / autoencoding shard / SVR shard keep 64 points per wave (4, 1, 1.25 rounds of 512)
    assert plan(4, 64, 2048, 37) == (64, 2048)
    assert plan(4, 16, 2048, 33) == (64, 512)
    assert plan(4, 16, 2500, 33) in ((64, 640), (32, 1280))     # 1.25 rounds of 512 large tiles or 1.67 of 768 small ones: measured equal
    assert plan(1, 32, 2048, 64) == (32, 512)                   # M1: f = 64 at 64 points per wave runs the generic body, 2x the cost
    # whole rounds: 24 x 2048 points are 768 small workgroups = exactly three per compute unit (or 192 large ones, one each);
    # 16 x 2048 are 256 middle ones, 8 x 2048 are 256 small ones -- never a round with one straggler
    assert plan(1, 24, 2048, 33) in ((16, 768), (64, 192))
    assert plan(1, 16, 2048, 33) == (32, 256) and plan(1, 8, 2048, 33) == (16, 256)
    # one shape split over 16 components (the reference's sampling call): the smallest tile, 2 workgroups per component
    segs = [(k * 128, (k + 1) * 128) for k in range(16)]
    assert plan(16, 1, 2048, 19, segments=segs) == (16, 32)
    # a forced tile is honoured per call, and the host-side context manager restores the word on exit (also on an exception)
    assert plan(4, 64, 2048, 37, word=32) == (32, 4096)
    with pytest.raises(RuntimeError):
        with _lib.tuning(points_per_wave=16):
            assert _lib.tune_word() == 16 and plan(4, 64, 2048, 37) == (16, 8192)
            raise RuntimeError('boom')
    assert _lib.tune_word() == 0
