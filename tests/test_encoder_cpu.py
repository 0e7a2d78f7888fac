"""Pin oracle/encoder_oracle.py to the genuine reference's encoder outputs (tests/golden/g11_encoder.npz) and check the
host-side modules (state_dict contract is asserted when the fixture is generated; here: heads on CPU, loud failure of the
point encoder without a HIP device)."""
import numpy as np
import pytest
import torch

from conftest import golden
from helpers import maxabs
from go_with_the_flows_amd import encoders, _lib
from go_with_the_flows_amd.synth import load_synth_
from oracle import encoder_oracle as eo

CASES = (('big', [128, 256, 512], 1100), ('small', [128, 64, 128], 1110))


@pytest.mark.parametrize('training', [False, True])
@pytest.mark.parametrize('tag,n_features,seed', CASES)
def test_pointnet_oracle_matches_reference(tag, n_features, seed, training):
    G = golden('g11_encoder')
    m = encoders.PointNetCloudEncoder(3, 64, n_features)
    st = load_synth_(m, seed)
    feat = eo.pointnet_features(G[f'{tag}_x'], st, len(n_features), training)
    t = 'train' if training else 'eval'
    scale = max(1.0, float(np.abs(G[f'{tag}_{t}_pooled']).max()))
    assert maxabs(feat.max(2), G[f'{tag}_{t}_pooled']) < 2e-5 * scale       # reference is fp32, oracle fp64
    assert maxabs(feat[:, :, :6], G[f'{tag}_{t}_feat_head']) < 2e-5 * scale


@pytest.mark.parametrize('training', [False, True])
def test_head_oracle_and_modules_match_reference(training):
    G = golden('g11_encoder')
    t = 'train' if training else 'eval'
    x = G['head_x']
    for tag, cls, det in (('post', encoders.FeatureEncoder, False), ('det', encoders.FeatureEncoder, True),
                          ('wts', encoders.WeightsEncoder, True)):
        m = cls(2, 48, 10, deterministic=det)
        st = load_synth_(m, 1130)
        m.train(training)
        with torch.no_grad():
            y = m(torch.from_numpy(x))
        ref = eo.feature_encoder(x, st, 2, det, training)
        if tag == 'post':
            for got, orc, key in ((y[0], ref[0], 'mu'), (y[1], ref[1], 'lv')):
                assert maxabs(got.numpy(), G[f'head_post_{t}_{key}']) < 1e-5
                assert maxabs(orc, G[f'head_post_{t}_{key}']) < 1e-5
        else:
            if tag == 'wts':
                ref = ref - np.log(np.exp(ref).sum(1, keepdims=True))
            assert maxabs(y.numpy(), G[f'head_{tag}_{t}']) < 1e-5
            assert maxabs(ref, G[f'head_{tag}_{t}']) < 1e-5


def test_point_encoder_refuses_cpu_tensors_and_unbuilt_shapes():
    m = encoders.PointNetCloudEncoder(3, 64, [128, 256, 512]).eval()
    with pytest.raises(_lib.GwtfError):
        m(torch.zeros(1, 3, 8))
    with pytest.raises(_lib.GwtfError):
        m.forward_max(torch.zeros(1, 3, 8))
    import ctypes
    w = (ctypes.c_int * 5)(3, 64, 128, 256, 512)
    L = _lib.lib()
    assert L.gwtf_encoder_raw_floats(w, 5) == sum(t.numel() for t in m._sources())
    assert L.gwtf_encoder_packed_floats(w, 5) == 1280 + 21 * 8192     # head (layer-0 table + biases, padded) + 21 chunks
    bad = (ctypes.c_int * 5)(3, 48, 128, 256, 512)
    assert L.gwtf_encoder_packed_floats(bad, 5) == 0
