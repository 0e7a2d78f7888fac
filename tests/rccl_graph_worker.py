"""Run by tests/test_gpu_models.py in a subprocess: the DATA-PARALLEL training step of the whole model as ONE hipGraph, with every
exchange captured inside it -- the packed BatchNorm-statistic all-reduces of the decoders' phase-split pipeline, the encoder's,
the row all-gathers of the per-shape modules and the overlapped gradient all-reduce -- over RCCL (backend 'nccl') on a 1-rank
group with GWTF_FORCE_SHARDED=1: the only multi-rank configuration one GPU allows (RCCL refuses two ranks on one device).
Checked against the plain single-process eager step: loss terms, every parameter gradient, every buffer, and the parameters after
the optimiser step (within Adam's +-lr)."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from go_with_the_flows_amd import autograd as gwa, models, optim                 # noqa: E402
from go_with_the_flows_amd.synth import load_synth_                            # noqa: E402
from go_with_the_flows_amd.training import GraphedTrainStep                    # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)


def build(sync):
    cfg = dict(json.load(open(os.path.join(GOLDEN, 'contract_model.json')))['cfg'], pc_enc_n_features=[128, 256, 512])
    m = models.Flow_Mixture_Model(**cfg)
    load_synth_(m, 1310)
    m = m.cuda().train()
    if sync:
        m = torch.nn.SyncBatchNorm.convert_sync_batchnorm(m)                   # train_ae.py:152
    return m, cfg


D = np.load(os.path.join(GOLDEN, 'g13_full_model.npz'))
g_in, p_in, noise = (torch.from_numpy(D[k]).cuda() for k in ('gcloud', 'pcloud', 'noise_g'))

# ---- the plain step: one process, no process group ----
m1, cfg = build(False)
m1.reparameterize = lambda mu, logvar: noise * torch.exp(0.5 * logvar) + mu
crit = models.Flow_Mixture_Loss(**cfg)
opt1 = optim.Adam(m1.parameters(), lr=1e-3, amsgrad=True)
terms1 = crit.fused(*m1.forward_fused(g_in, p_in))
terms1[0].backward()
grads1 = {n: q.grad.clone() for n, q in m1.named_parameters() if q.grad is not None}
opt1.step()
torch.cuda.synchronize()

# ---- the sharded step, graphed, over RCCL ----
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', str(29650 + os.getpid() % 200))
os.environ['GWTF_FORCE_SHARDED'] = '1'
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
m2, _ = build(True)
m2.reparameterize = lambda mu, logvar: noise * torch.exp(0.5 * logvar) + mu
opt2 = optim.Adam(m2.parameters(), lr=1e-3, amsgrad=True)
gwa.COLLECTIVES['n'] = 0
step = GraphedTrainStep(m2, crit, opt2, g_in, p_in, warmup_iters=2)
assert step.reducer is not None
C = 3 * m2.pc_decoder[0].n_flows
per_pass = 4 * C                                   # ONE packed statistic all-reduce per pipeline phase, all K components together
assert gwa.COLLECTIVES['n'] == 3 * per_pass, (gwa.COLLECTIVES, per_pass)       # 2 warm-up passes + the captured one
assert step.reducer.launched == 3 * len(m2.pc_decoder)                        # one async gradient all-reduce per decoder and pass
terms2 = step(g_in, p_in)                                                      # ONE replay + the optimiser step
torch.cuda.synchronize()
terms2 = [float(t) for t in terms2]                    # static tensors: the next replay overwrites them
rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-12))
worst = {'terms': max(abs(float(a) - float(b)) / max(1.0, abs(float(b))) for a, b in zip(terms2, terms1))}
# per tensor, relative to its own largest entry -- but no finer than 1e-4 of the whole gradient's (the top encoder BatchNorm's bias
# gradient is analytically ZERO: the posterior head's BatchNorm removes a per-channel shift; what is there is rounding noise)
gmax = max(float(v.abs().max()) for v in grads1.values())
per = {n: float((q.grad - grads1[n]).abs().max() / max(float(grads1[n].abs().max()), 1e-4 * gmax))
       for n, q in m2.named_parameters() if q.grad is not None}
worst['grads'] = max(per.values())
print('worst gradients:', sorted(per.items(), key=lambda kv: -kv[1])[:6], flush=True)
assert {n for n, q in m2.named_parameters() if q.grad is not None} == set(grads1)
sd1, sd2 = m1.state_dict(), m2.state_dict()
worst['buffers'] = max(rel(sd2[k].float(), sd1[k].float()) for k in sd1 if 'running' in k or 'num_batches' in k)
# after the optimiser step: Adam's first update is -lr * g / (|g| + eps) -- +-lr whatever |g| is, so entries whose gradient is rounding
# noise may land 2 lr apart; everything must be within that, and must have moved
worst['params'] = max(float((sd2[k] - sd1[k]).abs().max()) for k, _ in m1.named_parameters()) / 1e-3
print('GRAPH1', ' '.join(f'{k}={v:.2e}' for k, v in worst.items()), f'collectives_in_graph={per_pass}', flush=True)
# gradients: the statistic sums take a different route (phase-split, replicas summed in another order); 1e-3 of each tensor's max
assert worst['terms'] < 1e-5 and worst['grads'] < 2e-3 and worst['buffers'] < 1e-4 and worst['params'] < 2.2, worst
# replays keep training: the loss goes down over a few steps
losses = [float(step(g_in, p_in)[0]) for _ in range(6)]
assert all(np.isfinite(losses)) and losses[-1] < terms2[0], (terms2[0], losses)
dist.barrier()
dist.destroy_process_group()
print('GRAPH1 ok', flush=True)
