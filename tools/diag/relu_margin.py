"""Smallest |pre-activation| of the first layer of every coupling branch (train-mode BatchNorm): how close is the nearest ReLU to
its kink on the contract model's inputs?"""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import json, os, numpy as np, torch
from go_with_the_flows_amd import models
from go_with_the_flows_amd.synth import load_synth_
G = 'tests/golden'
D = np.load(os.path.join(G, 'g13_full_model.npz'))
dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
base = sys.argv[1] if len(sys.argv) > 1 else 'freevar'
cfg = dict(json.load(open(os.path.join(G, 'contract_model.json')))['cfg'], p_decoder_base_type=base)
m = models.Flow_Mixture_Model(**cfg); load_synth_(m, 1310); m = m.cuda().train()
noise = dev(D['noise_g']); m.reparameterize = lambda mu, lv: noise * torch.exp(0.5 * lv) + mu
gcloud, pcloud = dev(D['gcloud']), dev(D['pcloud'])
with torch.no_grad():
    enc = m.encode(gcloud)
    gs = enc['g_posterior_samples']
    z, ld, (ps, mus, lvs) = m.mixture_stack().forward_all_lists(pcloud, gs, mode='inverse')
print('ps', tuple(ps.shape))
for k, d in enumerate(m.pc_decoder):
    c = 0
    for fl in d.flows:
        for name in ('nvp1', 'nvp2', 'nvp3'):
            cp = getattr(fl, name)
            for br in ('mu', 'logvar'):
                blk = getattr(cp, 'T_%s_0' % br)
                w = getattr(blk, '%s_sd0' % br).weight.double()[0]              # (f, in)
                bn = getattr(blk, '%s_sd0_bn' % br)
                best = None
                # the conditioning coordinates: every subset of matching size, the smallest margin is reported per subset
                import itertools
                for idx in itertools.combinations(range(3), w.shape[1]):
                    for slot in range(ps.shape[1]):
                        x = ps[k, slot].double()[:, list(idx), :]                # (B, in, N)
                        a = torch.einsum('fi,bin->bfn', w, x)
                        mu_, var = a.mean((0, 2), keepdim=True), a.var((0, 2), unbiased=False, keepdim=True)
                        pre = bn.weight.double()[None, :, None] * (a - mu_) / torch.sqrt(var + bn.eps) + bn.bias.double()[None, :, None]
                        v = float(pre.abs().min())
                        if best is None or v < best[0]:
                            best = (v, idx, slot)
                print('dec %d coupling %d %-6s min|pre| %.2e (coords %s, list slot %d)' % (k, c, br, *best))
            c += 1
