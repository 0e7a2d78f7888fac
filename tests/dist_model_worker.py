"""Worker for tests/test_gpu_models.py::test_two_rank_whole_model_training_step_matches_single_process.
Two ranks share cuda:0 (one-GPU box), gloo carries the collectives.  The whole Flow_Mixture_Model is converted to SyncBatchNorm
as the reference does (train_ae.py:152); each rank owns half of the batch.  Loss, the parameter gradients averaged over the ranks
(DDP semantics, train_ae.py:153) and every running statistic must equal the single-process full-batch step."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from go_with_the_flows_amd import models                      # noqa: E402
from go_with_the_flows_amd.dist import shard_bounds, all_reduce_gradients   # noqa: E402
from go_with_the_flows_amd.synth import load_synth_           # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def build():
    cfg = dict(json.load(open(os.path.join(GOLDEN, 'contract_model.json')))['cfg'], pc_enc_n_features=[128, 256, 512])
    m = models.Flow_Mixture_Model(**cfg)
    load_synth_(m, 1310)
    return m.cuda().train(), cfg


def step(m, cfg, g_in, p_in, noise):
    m.reparameterize = lambda mu, logvar: noise * torch.exp(0.5 * logvar) + mu
    crit = models.Flow_Mixture_Loss(**cfg)
    enc, dec = m.forward_fused(g_in, p_in)
    loss = crit.fused(enc, dec)[0]
    loss.backward()
    return loss


def flat_state(m):
    return torch.cat([v.reshape(-1).float() for k, v in m.state_dict().items() if 'running' in k]).cpu().numpy()


def main():
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    D = np.load(os.path.join(GOLDEN, 'g13_full_model.npz'))
    g_all, p_all, noise = (torch.from_numpy(D[k]).cuda() for k in ('gcloud', 'pcloud', 'noise_g'))
    rows = int(os.environ.get('GWTF_ROWS', '0'))
    if rows:
        # a batch beyond 128 shapes (e.g. 96 per rank): every per-shape module sees the gathered rows of both ranks, which the HIP
        # kernels walk in row blocks (csrc/gwtf_heads.hip, gwtf_prior.hip, gwtf_film_train.hip) -- no library fallback
        rng = np.random.default_rng(4242)
        mk = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).cuda()
        g_all, p_all, noise = 0.5 * mk(rows, 3, g_all.shape[2]), 0.5 * mk(rows, 3, p_all.shape[2]), mk(rows, noise.shape[1])
    B = g_all.shape[0]
    b0, b1 = shard_bounds(B, rank, world)
    m, cfg = build()
    m = torch.nn.SyncBatchNorm.convert_sync_batchnorm(m)
    if rows:
        assert m.g_prior._fused_ok(torch.zeros(b1 - b0, cfg['g_latent_space_size'], device='cuda'), rows)
        assert m.g_posterior._hip_layers(torch.zeros(rows, cfg['pc_enc_n_features'][-1], device='cuda')) is not None
    loss = step(m, cfg, g_all[b0:b1], p_all[b0:b1], noise[b0:b1])
    all_reduce_gradients(m, average=True)
    grads = torch.cat([q.grad.reshape(-1) for q in m.parameters() if q.grad is not None])
    lt = loss.detach().clone()
    dist.all_reduce(lt)
    np.savez(os.path.join(os.environ['GWTF_TMP'], f'model{rank}.npz'), grads=grads.cpu().numpy(), loss=float(lt) / world,
             state=flat_state(m))
    # the reference's own wrapping and call (train_ae.py:152-153, training.py:43-54): SyncBatchNorm + DistributedDataParallel around
    # the model, model(g, p) -> lists -> criterion -> backward; DDP's averaged gradients must equal the flat all-reduce's
    m2, cfg2 = build()
    m2 = torch.nn.SyncBatchNorm.convert_sync_batchnorm(m2)
    m2.reparameterize = lambda mu, logvar: noise[b0:b1] * torch.exp(0.5 * logvar) + mu
    ddp = torch.nn.parallel.DistributedDataParallel(m2, find_unused_parameters=True)
    crit2 = models.Flow_Mixture_Loss(**cfg2)
    output_prior, output_decoder, logits = ddp(g_all[b0:b1], p_all[b0:b1])
    crit2(output_prior, output_decoder, logits)[0].backward()
    ddp_grads = torch.cat([q.grad.reshape(-1) for q in m2.parameters() if q.grad is not None])
    ddp_err = float((ddp_grads - grads).abs().max() / grads.abs().max())
    assert ddp_grads.numel() == grads.numel() and ddp_err < 1e-5, ddp_err
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        m1, cfg = build()
        loss1 = step(m1, cfg, g_all, p_all, noise)
        g1 = torch.cat([q.grad.reshape(-1) for q in m1.parameters() if q.grad is not None]).cpu().numpy()
        part = np.load(os.path.join(os.environ['GWTF_TMP'], 'model0.npz'))
        rel = lambda a, b: float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))
        res = {'loss': abs(float(part['loss']) - float(loss1)) / abs(float(loss1)), 'grads': rel(part['grads'], g1),
               'running': rel(part['state'], flat_state(m1))}
        # per top-level module, for the failure message
        off, per = 0, {}
        for name, q in m1.named_parameters():
            if q.grad is None:
                continue
            n = q.numel()
            a, b = part['grads'][off:off + n], g1[off:off + n]
            key = name.split('.')[0]
            per[key] = max(per.get(key, 0.0), float(np.abs(a - b).max() / (np.abs(g1).max() + 1e-12)))
            off += n
        print('DDP list-API step == flat all-reduce step', flush=True)
        print('MODEL2', ' '.join(f'{k}={v:.2e}' for k, v in res.items()), '|', ' '.join(f'{k}={v:.1e}' for k, v in per.items()), flush=True)
        assert res['loss'] < 1e-5 and res['grads'] < 2e-3 and res['running'] < 1e-4, res


if __name__ == '__main__':
    main()
