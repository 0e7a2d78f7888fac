"""Reproducibility of train-mode gradients of ONE decoder: list-slot losses vs fused outputs, twice each."""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import decoder_and_state
from go_with_the_flows_amd.synth import synth_inputs
L, f, G, B, N = 2, 8, 16, 4, 48
p, g = synth_inputs(B, N, G, 1)
def run(kind):
    m, _ = decoder_and_state(L, f, G, 7); m = m.cuda().train()
    pd, gd = torch.from_numpy(p).cuda().requires_grad_(True), torch.from_numpy(g).cuda().requires_grad_(True)
    if kind == 'fused':
        z, ld = m.forward_fused(pd, gd, 'inverse'); loss = (z * z).sum() + ld.sum()
    elif kind == 'lists_all':
        ps, mus, lvs = m(pd, gd, 'inverse'); loss = (ps[0] * ps[0]).sum() + sum(lvs).sum()
    elif kind == 'lists_lv_only':
        ps, mus, lvs = m(pd, gd, 'inverse'); loss = sum(lvs).sum()
    elif kind == 'lists_ps_only':
        ps, mus, lvs = m(pd, gd, 'inverse'); loss = (ps[0] * ps[0]).sum()
    loss.backward()
    return float(loss), {n: q.grad.clone() for n, q in m.named_parameters()}, pd.grad.clone()
for kind in ('fused', 'lists_all', 'lists_lv_only', 'lists_ps_only'):
    a, b = run(kind), run(kind)
    worst = max(((float((a[1][n] - b[1][n]).abs().max() / (b[1][n].abs().max() + 1e-3)), n) for n in a[1]))
    print(kind, 'loss', a[0], b[0], 'worst param', worst, 'dp', float((a[2] - b[2]).abs().max()))
fa, la = run('fused'), run('lists_all')
worst = max(((float((fa[1][n] - la[1][n]).abs().max() / (la[1][n].abs().max() + 1e-3)), n) for n in fa[1]))
print('fused vs lists_all', worst, float((fa[2] - la[2]).abs().max()))
