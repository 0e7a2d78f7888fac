"""Run-to-run deviation of the whole-model training gradients (fused step, state restored every time).
usage: run_to_run_noise.py [base_type] [jitter_seed]   -- jitter_seed adds 1e-3 * randn to the clouds (0 = the golden clouds)

Finding (docs/LOG.md 4.11): with the golden clouds and base type 'freevar', one first-layer pre-activation of decoder 0 / coupling 1 /
logvar branch sits 1.3e-7 from the ReLU kink (tools/diag/relu_margin.py); the batch statistics are summed with float atomics, so
the value lands on either side from run to run and the whole gradient takes one of TWO values (8.7e-5 of its norm apart)."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import json, os, numpy as np, torch
from go_with_the_flows_amd import models
from go_with_the_flows_amd.synth import load_synth_
G = 'tests/golden'
D = np.load(os.path.join(G, 'g13_full_model.npz'))
dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
base = sys.argv[1] if len(sys.argv) > 1 else 'freevar'
jitter = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cfg = dict(json.load(open(os.path.join(G, 'contract_model.json')))['cfg'], p_decoder_base_type=base)
m = models.Flow_Mixture_Model(**cfg); load_synth_(m, 1310); m = m.cuda().train()
noise = dev(D['noise_g']); m.reparameterize = lambda mu, lv: noise * torch.exp(0.5 * lv) + mu
crit = models.Flow_Mixture_Loss(**cfg)
state = {k: v.clone() for k, v in m.state_dict().items()}
gcloud, pcloud = dev(D['gcloud']), dev(D['pcloud'])
if jitter:
    gen = torch.Generator().manual_seed(jitter)
    gcloud = gcloud + 1e-3 * torch.randn(gcloud.shape, generator=gen).cuda()
    pcloud = pcloud + 1e-3 * torch.randn(pcloud.shape, generator=gen).cuda()
runs = []
for rep in range(40):
    m.load_state_dict(state); m.zero_grad(set_to_none=True)
    enc, dec = m.forward_fused(gcloud, pcloud)
    crit.fused(enc, dec)[0].backward()
    runs.append(torch.cat([p.grad.reshape(-1) for p in m.parameters() if p.grad is not None]).clone())
d = [float((r - runs[0]).norm() / runs[0].norm()) for r in runs[1:]]
print(base, 'jitter', jitter, 'deviating runs (> 1e-5 of the gradient norm):', sum(x > 1e-5 for x in d), 'of', len(d), 'max %.1e' % max(d))
