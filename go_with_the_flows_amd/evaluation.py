"""Generation / reconstruction metrics built on the structural-loss kernels (metrics.py).

Host-side mirror of what lib/networks/evaluating.py imports: lib/metrics/evaluation_metrics.py (distChamfer :33-45,
EMD_CD_F1 :47-107, _pairwise_EMD_CD_F1_SCORE :110-181, knn :185-218, lgan_mmd_cov :220-239, compute_all_metrics :242-329)
and lib/networks/utils.py (get_voxel_occ_dist :45-79, JSD :82-86).  Same names, arguments, return keys and conventions
(which output is "left" and which "right", 1-based vs 0-based, biased means), so evaluate_ae.py runs unchanged on ROCm.
The O(N^2) work per cloud pair runs in csrc/gwtf_metrics.hip; everything here is orchestration.
"""
import numpy as np
import torch
from scipy.stats import entropy

from .metrics import distChamferCUDA, emd_approx, f_score  # noqa: F401  (re-exported: evaluating.py imports them from here)


def distChamfer(a, b):
    """Pure-torch Chamfer via |x|^2 + |y|^2 - 2 x.y (reference :33-45).  NOTE the reference's return order: first the
    distance of every point of ``b`` to its nearest in ``a`` (min over dim 1), then the other direction; a and b must
    have the same number of points (its diagonal indexing assumes so)."""
    x, y = a, b
    xx = (x * x).sum(2)                       # (B, N)
    yy = (y * y).sum(2)
    P = xx.unsqueeze(2) + yy.unsqueeze(1) - 2.0 * torch.bmm(x, y.transpose(2, 1))
    return P.min(1)[0], P.min(2)[0]


def _cd_pair(x, y, accelerated_cd):
    return distChamferCUDA(x, y) if accelerated_cd else distChamfer(x, y)


def _f1(dl, dr, threshold):
    precision = 100. * (dr < threshold).float().mean(1)
    recall = 100. * (dl < threshold).float().mean(1)
    return 2. * precision * recall / (precision + recall + 1e-7)


def EMD_CD_F1(sample_pcs, ref_pcs, batch_size, accelerated_cd=False, reduced=True, cd_option=False, emd_option=False,
              one_part_of_cd=False, f1_option=False, f1_threshold=0.0001):
    """Cloud i of ``sample_pcs`` against cloud i of ``ref_pcs`` (reference :47-107).  Options that are off report 0."""
    n = sample_pcs.shape[0]
    assert n == ref_pcs.shape[0], "REF:%d SMP:%d" % (ref_pcs.shape[0], n)
    acc = {'CD': [], 'EMD': [], 'F1': [], 'CDL': [], 'CDR': []}
    for lo in range(0, n, batch_size):
        s, r = sample_pcs[lo:lo + batch_size], ref_pcs[lo:lo + batch_size]
        dl, dr = _cd_pair(s, r, accelerated_cd)
        if cd_option:
            acc['CD'].append(dl.mean(dim=1) + dr.mean(dim=1))
        if one_part_of_cd:
            acc['CDL'].append(dl.mean(dim=1))
            acc['CDR'].append(dr.mean(dim=1))
        if emd_option:
            acc['EMD'].append(emd_approx(s, r))
        if f1_option:
            acc['F1'].append(_f1(dl, dr, f1_threshold))
    out = {}
    for key, vals in acc.items():
        if vals:
            v = torch.cat(vals)
            out[key] = v.mean() if reduced else v
        else:
            out[key] = 0
    return out


def _pairwise_EMD_CD_F1_SCORE(sample_pcs, ref_pcs, batch_size, f1_threshold, accelerated_cd=True, cd_option=False,
                              one_part_of_cd=False, emd_option=False, f1_option=False):
    """Every sample cloud against every reference cloud -> (N_sample, N_ref) matrices (reference :110-181); a matrix whose
    option is off comes back as an empty list, as in the reference."""
    rows = {'cd': [], 'emd': [], 'f1': [], 'left': [], 'right': []}
    n_ref = ref_pcs.shape[0]
    for i in range(sample_pcs.shape[0]):
        cur = {k: [] for k in rows}
        for lo in range(0, n_ref, batch_size):
            ref = ref_pcs[lo:lo + batch_size]
            smp = sample_pcs[i].view(1, -1, 3).expand(ref.size(0), -1, -1).contiguous()
            dl, dr = _cd_pair(smp, ref, accelerated_cd)
            if one_part_of_cd:
                cur['left'].append(dl.mean(dim=1).view(1, -1))
                cur['right'].append(dr.mean(dim=1).view(1, -1))
            if cd_option:
                cur['cd'].append((dl.mean(dim=1) + dr.mean(dim=1)).view(1, -1))
            if emd_option:
                cur['emd'].append(emd_approx(smp, ref).view(1, -1))
            if f1_option:
                cur['f1'].append(_f1(dl, dr, f1_threshold).view(1, -1))
        for k in rows:
            if cur[k]:
                rows[k].append(torch.cat(cur[k], dim=1))
    mat = {k: (torch.cat(v, dim=0) if v else []) for k, v in rows.items()}
    return mat['cd'], mat['emd'], mat['f1'], mat['left'], mat['right']


def knn(Mxx, Mxy, Myy, k, sqrt=False):
    """Leave-one-out k-NN two-sample test on the joint distance matrix (reference :185-218, after Xu et al.)."""
    n0, n1 = Mxx.size(0), Myy.size(0)
    label = torch.cat((torch.ones(n0), torch.zeros(n1))).to(Mxx)
    M = torch.cat((torch.cat((Mxx, Mxy), 1), torch.cat((Mxy.transpose(0, 1), Myy), 1)), 0)
    if sqrt:
        M = M.abs().sqrt()
    _, idx = (M + torch.diag(float('inf') * torch.ones(n0 + n1).to(Mxx))).topk(k, 0, False)
    count = torch.zeros(n0 + n1).to(Mxx)
    for i in range(k):
        count = count + label.index_select(0, idx[i])
    pred = torch.ge(count, (float(k) / 2) * torch.ones(n0 + n1).to(Mxx)).float()
    s = {'tp': (pred * label).sum(), 'fp': (pred * (1 - label)).sum(),
         'fn': ((1 - pred) * label).sum(), 'tn': ((1 - pred) * (1 - label)).sum()}
    s.update({'precision': s['tp'] / (s['tp'] + s['fp'] + 1e-10), 'recall': s['tp'] / (s['tp'] + s['fn'] + 1e-10),
              'acc_t': s['tp'] / (s['tp'] + s['fn'] + 1e-10), 'acc_f': s['tn'] / (s['tn'] + s['fp'] + 1e-10),
              'acc': torch.eq(label, pred).float().mean()})
    return s


def lgan_mmd_cov(all_dist, mode='min'):
    """Minimum matching distance and coverage from an (N_sample, N_ref) matrix (reference :220-239)."""
    n_ref = all_dist.size(1)
    pick = torch.min if mode == 'min' else torch.max
    if mode not in ('min', 'max'):
        raise ValueError(f"mode must be 'min' or 'max', got {mode!r}")
    val_fromsmp, idx = pick(all_dist, dim=1)
    val, idx_mmd = pick(all_dist, dim=0)
    cov = torch.tensor(float(idx.unique().view(-1).size(0)) / float(n_ref)).to(all_dist)
    return {'lgan_mmd': val.mean(), 'lgan_cov': cov, 'lgan_mmd_smp': val_fromsmp.mean(), 'idx_mmd': idx_mmd,
            'mmd_contrib': val}


def compute_all_metrics(sample_pcs, ref_pcs, batch_size, accelerated_cd=False, f1_threshold=0.001, cd_option=False,
                        one_part_of_cd=False, emd_option=False, f1_option=False):
    """MMD / COV / 1-NN accuracy for every enabled distance (reference :242-329)."""
    opts = dict(accelerated_cd=accelerated_cd, f1_threshold=f1_threshold, cd_option=cd_option,
                one_part_of_cd=one_part_of_cd, emd_option=emd_option, f1_option=f1_option)
    names = ('CD', 'EMD', 'F1', 'CD-left', 'CD-right')
    enabled = (cd_option, emd_option, f1_option, one_part_of_cd, one_part_of_cd)
    rs = _pairwise_EMD_CD_F1_SCORE(sample_pcs, ref_pcs, batch_size, **opts)
    results = {}
    for name, on, M in zip(names, enabled, rs):
        if on:
            res = lgan_mmd_cov(M, 'max' if name == 'F1' else 'min')
            results.update({'%s-%s' % (k, name): v for k, v in res.items()})
    rr = _pairwise_EMD_CD_F1_SCORE(ref_pcs, ref_pcs, batch_size, **opts)
    ss = _pairwise_EMD_CD_F1_SCORE(sample_pcs, sample_pcs, batch_size, **opts)
    for name, on, M_rs, M_rr, M_ss in zip(names, enabled, rs, rr, ss):
        if on:
            one_nn = knn(M_ss, M_rs, M_rr, 1, sqrt=False)
            results.update({'1-NN-%s-%s' % (name, k): v for k, v in one_nn.items() if 'acc' in k})
    return results


def get_voxel_occ_dist(all_clouds, clouds_flag='gen', res=28, bound=0.5, bs=128, warning=True):
    """Occupancy histogram of all points over a res^3 grid of [-0.5, 0.5)^3, normalised (reference utils.py:45-79); points
    outside the cube are dropped.  ``bs`` is accepted for signature compatibility (the reference chunks by it)."""
    if np.any(np.fabs(all_clouds) > bound) and warning:
        print('{} clouds out of cube bounds: [-{}; {}]'.format(clouds_flag, bound, bound))
    n_nans = np.isnan(all_clouds).sum()
    if n_nans > 0:
        print('{} NaN values in point cloud tensors.'.format(n_nans))
    edges = -0.5 + np.arange(res + 1) * (1. / res)
    pts = np.asarray(all_clouds).reshape(-1, 3)
    idx = np.stack([np.searchsorted(edges, pts[:, d], side='right') - 1 for d in range(3)], axis=1)   # edges[i] <= x < edges[i+1]
    ok = np.all((idx >= 0) & (idx < res), axis=1)
    hist = np.zeros((res, res, res), dtype=np.uint64)
    np.add.at(hist, tuple(idx[ok].T), np.uint64(1))
    return np.float64(hist) / hist.sum()


def JSD(clouds1, clouds2, clouds1_flag='gen', clouds2_flag='ref', warning=True):
    """Jensen-Shannon divergence (base 2) between the two sets' occupancy histograms (reference utils.py:82-86)."""
    d1 = get_voxel_occ_dist(clouds1, clouds_flag=clouds1_flag, warning=warning)
    d2 = get_voxel_occ_dist(clouds2, clouds_flag=clouds2_flag, warning=warning)
    return entropy((d1 + d2).flatten() / 2.0, base=2) - 0.5 * (entropy(d1.flatten(), base=2) + entropy(d2.flatten(), base=2))
