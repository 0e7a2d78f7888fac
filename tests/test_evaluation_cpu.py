"""Host logic of the evaluation metrics (go_with_the_flows_amd/evaluation.py) on CPU: the pure-torch Chamfer path, MMD /
coverage / 1-NN bookkeeping on hand-made matrices, and the occupancy-grid JSD against a direct restatement of the
reference's binning rule (lib/networks/utils.py:45-86: bin i holds -0.5 + i/res <= x < -0.5 + (i+1)/res, points outside
the cube are dropped)."""
import numpy as np
import torch
from scipy.stats import entropy

from go_with_the_flows_amd import evaluation as ev


def test_dist_chamfer_matches_bruteforce_and_reference_return_order():
    g = torch.Generator().manual_seed(0)
    a, b = torch.randn(3, 40, 3, generator=g), torch.randn(3, 40, 3, generator=g)
    first, second = ev.distChamfer(a, b)
    D = ((a[:, :, None, :] - b[:, None, :, :]) ** 2).sum(-1)        # (B, n_a, n_b)
    assert torch.allclose(first, D.min(1)[0], atol=1e-5)           # per point of b: nearest in a
    assert torch.allclose(second, D.min(2)[0], atol=1e-5)


def test_emd_cd_f1_cpu_path_reduction_and_defaults():
    g = torch.Generator().manual_seed(1)
    s, r = torch.rand(5, 30, 3, generator=g) - 0.5, torch.rand(5, 30, 3, generator=g) - 0.5
    out = ev.EMD_CD_F1(s, r, batch_size=2, accelerated_cd=False, reduced=False, cd_option=True, one_part_of_cd=True,
                       f1_option=True, f1_threshold=0.01)
    dl, dr = ev.distChamfer(s, r)
    assert torch.allclose(out['CD'], dl.mean(1) + dr.mean(1)) and out['CD'].shape == (5,)
    assert torch.allclose(out['CDL'], dl.mean(1)) and torch.allclose(out['CDR'], dr.mean(1))
    p, q = 100. * (dr < 0.01).float().mean(1), 100. * (dl < 0.01).float().mean(1)
    assert torch.allclose(out['F1'], 2 * p * q / (p + q + 1e-7))
    assert out['EMD'] == 0                                           # option off -> the reference's initial value
    red = ev.EMD_CD_F1(s, r, batch_size=5, cd_option=True)
    assert torch.allclose(red['CD'], out['CD'].mean())


def test_lgan_mmd_cov_and_knn_on_known_matrices():
    M = torch.tensor([[0.1, 0.9, 0.8], [0.7, 0.2, 0.6]])           # 2 samples x 3 refs
    res = ev.lgan_mmd_cov(M)
    assert abs(float(res['lgan_mmd']) - (0.1 + 0.2 + 0.6) / 3) < 1e-6       # per ref: nearest sample
    assert abs(float(res['lgan_mmd_smp']) - 0.15) < 1e-6
    assert abs(float(res['lgan_cov']) - 2 / 3) < 1e-6                        # refs 0 and 1 are somebody's nearest
    assert res['idx_mmd'].tolist() == [0, 1, 1]
    assert abs(float(ev.lgan_mmd_cov(M, 'max')['lgan_mmd']) - (0.7 + 0.9 + 0.8) / 3) < 1e-6
    # two well separated clusters: 1-NN classifies everything correctly; identical sets: chance or worse
    far = torch.full((3, 3), 10.0)
    near = torch.rand(3, 3) * 0.1
    near = (near + near.t()) / 2
    assert float(ev.knn(near, far, near, 1)['acc']) == 1.0
    s = ev.knn(near, near, near, 1)
    assert 0.0 <= float(s['acc']) <= 1.0 and set(s) >= {'tp', 'fp', 'fn', 'tn', 'precision', 'recall', 'acc_t', 'acc_f', 'acc'}


def test_compute_all_metrics_cpu_path_keys_and_self_comparison():
    g = torch.Generator().manual_seed(2)
    a = torch.rand(6, 20, 3, generator=g) - 0.5
    res = ev.compute_all_metrics(a, a.clone(), batch_size=4, accelerated_cd=False, cd_option=True, one_part_of_cd=True,
                                 f1_option=True, f1_threshold=0.001)
    for key in ('lgan_mmd-CD', 'lgan_cov-CD', 'lgan_mmd_smp-CD', '1-NN-CD-acc', '1-NN-CD-acc_t', '1-NN-CD-acc_f',
                'lgan_mmd-F1', 'lgan_cov-CD-left', '1-NN-CD-right-acc'):
        assert key in res, key
    assert float(res['lgan_mmd-CD']) < 1e-6 and float(res['lgan_cov-CD']) == 1.0      # every cloud matches itself


def _reference_rule_hist(clouds, res):
    edges = -0.5 + np.arange(res + 1) * (1. / res)
    h = np.zeros((res, res, res))
    for pt in clouds.reshape(-1, 3):
        ijk = []
        for d in range(3):
            inside = np.logical_and(edges[:-1] <= pt[d], pt[d] < edges[1:])
            ijk.append(int(inside.argmax()) if inside.any() else None)
        if None not in ijk:
            h[tuple(ijk)] += 1
    return h / h.sum()


def test_jsd_follows_the_reference_binning_rule():
    rng = np.random.default_rng(3)
    c1 = (rng.random((4, 50, 3)) - 0.5).astype(np.float32)
    c2 = (rng.random((3, 60, 3)) * 0.6 - 0.3).astype(np.float32)
    c1[0, 0] = [0.5, 0.0, 0.0]            # on the upper face: outside [-0.5, 0.5)
    c1[0, 1] = [-0.5, 0.2, -0.1]          # on the lower face: inside
    c1[0, 2] = [0.7, 0.0, 0.0]            # outside the cube: dropped
    for res in (28, 8):
        d1 = ev.get_voxel_occ_dist(c1, res=res, warning=False)
        assert np.allclose(d1, _reference_rule_hist(c1, res)) and abs(d1.sum() - 1) < 1e-12
    h1, h2 = _reference_rule_hist(c1, 28), _reference_rule_hist(c2, 28)
    want = entropy((h1 + h2).flatten() / 2, base=2) - 0.5 * (entropy(h1.flatten(), base=2) + entropy(h2.flatten(), base=2))
    assert abs(ev.JSD(c1, c2, warning=False) - want) < 1e-12
    assert ev.JSD(c1, c1, warning=False) < 1e-12


def _g14():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), 'golden', 'g14_evaluation.npz'))


def _close(a, b, tol=1e-6):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    return np.allclose(a, b, rtol=tol, atol=tol)


def test_g14_host_metrics_match_the_reference():
    """Vectors produced by the reference's own functions (lib/metrics/evaluation_metrics.py, lib/networks/utils.py), on every
    path that does not enter its CUDA extension (tests/golden/make_golden.py::g14_evaluation_metrics)."""
    D = _g14()
    T = torch.from_numpy
    first, second = ev.distChamfer(T(D['a']), T(D['b']))
    assert _close(first, D['chamfer_first'], 1e-5) and _close(second, D['chamfer_second'], 1e-5)
    smp, ref = T(D['smp']), T(D['ref'])
    r = ev.EMD_CD_F1(smp, ref, 4, accelerated_cd=False, reduced=False, cd_option=True, one_part_of_cd=True, f1_option=True,
                     f1_threshold=0.01)
    for ours, theirs in (('CD', 'pair_CD'), ('CDL', 'pair_CDL'), ('CDR', 'pair_CDR'), ('F1', 'pair_F1')):
        assert _close(r[ours], D[theirs], 1e-5), ours
    r = ev.EMD_CD_F1(smp, ref, 4, accelerated_cd=False, reduced=True, cd_option=True, f1_option=True, f1_threshold=0.01)
    assert _close(r['CD'], D['pair_CD_mean'], 1e-5) and _close(r['F1'], D['pair_F1_mean'], 1e-4)
    Mxx, Mxy, Myy = T(D['Mxx']), T(D['Mxy']), T(D['Myy'])
    for k, sq in ((1, False), (3, True)):
        res = ev.knn(Mxx, Mxy, Myy, k, sqrt=sq)
        for key, v in res.items():
            assert _close(v, D[f'knn{k}_{key}']), (k, key)
    for mode in ('min', 'max'):
        res = ev.lgan_mmd_cov(T(D['lgan_in']), mode)
        for key, v in res.items():
            assert _close(v, D[f'lgan_{mode}_{key}']), (mode, key)
    res = ev.compute_all_metrics(smp, ref[:5], 4, accelerated_cd=False, f1_threshold=0.01, cd_option=True, one_part_of_cd=True,
                                 f1_option=True, emd_option=False)
    assert sorted(res) == list(D['all_names'])
    for key in res:
        assert _close(res[key], D['all_' + key], 1e-5), key
    assert np.array_equal(ev.get_voxel_occ_dist(D['c1'], warning=False), D['occ1'])
    assert np.array_equal(ev.get_voxel_occ_dist(D['c2'], warning=False), D['occ2'])
    assert abs(ev.JSD(D['c1'], D['c2'], warning=False) - float(D['jsd'])) < 1e-12
