"""Worker for tests/test_gpu_parity.py::test_two_rank_syncbn_training_matches_single_process.
Two ranks share cuda:0 (one-GPU box), gloo carries the small statistic tensors.  Each rank owns half of the batch;
forward outputs, input gradients and (summed) parameter gradients must equal the single-process full-batch run."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import decoder_and_state                       # noqa: E402
from go_with_the_flows_amd.dist import shard_bounds          # noqa: E402
from go_with_the_flows_amd.synth import synth_inputs         # noqa: E402


def main():
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    L, f, G, B, N = 1, 8, 8, 5, 48
    p, g = synth_inputs(B, N, G, 11)
    rng = np.random.default_rng(12)
    wz, wl = rng.normal(size=(B, 3, N)).astype(np.float32), rng.normal(size=(B, 3, N)).astype(np.float32)
    b0, b1 = shard_bounds(B, rank, world)
    m, _ = decoder_and_state(L, f, G, 13)
    m = m.cuda().train()
    pt = torch.from_numpy(p[b0:b1]).cuda().requires_grad_(True)
    gt = torch.from_numpy(g[b0:b1]).cuda().requires_grad_(True)
    from go_with_the_flows_amd import autograd as gwa
    gwa.COLLECTIVES['n'] = 0
    z, ld = m.forward_fused(pt, gt, 'inverse')
    loss = (z * torch.from_numpy(wz[b0:b1]).cuda()).sum() + (ld * torch.from_numpy(wl[b0:b1]).cuda()).sum()
    loss.backward()
    # ONE packed statistic all-reduce per phase: 2 per depth level going forward (the level-0 moments + C statistics passes +
    # C-1 output moments) and 2 per level going backward -- independent of the number of mixture components
    C = 3 * L
    assert gwa.COLLECTIVES['n'] == 4 * C, gwa.COLLECTIVES
    # K = 2 mixture components through the same pipeline: still 4 C collectives; results == single process (checked below)
    import go_with_the_flows_amd as gw
    decs = [decoder_and_state(L, f, G, 20 + k)[0].cuda().train() for k in range(2)]
    ms = gw.MixtureStack(decs)
    ptm = torch.from_numpy(p[b0:b1]).cuda().requires_grad_(True)
    gtm = torch.from_numpy(g[b0:b1]).cuda().requires_grad_(True)
    gwa.COLLECTIVES['n'] = 0
    zm, ldm = ms.forward_all(ptm, gtm, 'inverse')
    wk = torch.tensor([1.0, -0.5], device='cuda').view(2, 1, 1, 1)
    ((zm * wk * torch.from_numpy(wz[b0:b1]).cuda()).sum() + (ldm * wk * torch.from_numpy(wl[b0:b1]).cuda()).sum()).backward()
    assert gwa.COLLECTIVES['n'] == 4 * C, gwa.COLLECTIVES
    from go_with_the_flows_amd.dist import all_reduce_gradients
    for d in decs:
        all_reduce_gradients(d, average=False)
    # the same gradients with the exchange overlapped: the decoders' flat gradient is all-reduced asynchronously from INSIDE
    # the backward pass, the remaining parameters afterwards
    from go_with_the_flows_amd.dist import OverlappedGradients
    holder = torch.nn.ModuleList(decs + [torch.nn.Linear(3, 2).cuda()])
    ref_grads = torch.cat([q.grad.reshape(-1) for d in decs for q in d.parameters()]).clone()
    dpm, dgm = ptm.grad.cpu().numpy().copy(), gtm.grad.cpu().numpy().copy()
    for q in holder.parameters():
        q.grad = None
    reducer = OverlappedGradients(holder, average=False)
    with reducer:
        zo, ldo = ms.forward_all(ptm, gtm, 'inverse')
        extra = holder[-1](torch.ones(1, 3, device='cuda')).sum() * (rank + 1.0)
        ((zo * wk * torch.from_numpy(wz[b0:b1]).cuda()).sum() + (ldo * wk * torch.from_numpy(wl[b0:b1]).cuda()).sum() + extra).backward()
    assert reducer.launched == 2                  # one flat gradient per decoder, launched from inside the backward pass
    got = torch.cat([q.grad.reshape(-1) for d in decs for q in d.parameters()])
    assert float((got - ref_grads).abs().max() / ref_grads.abs().max()) < 1e-5
    assert abs(float(holder[-1].bias.grad[0]) - sum(r + 1.0 for r in range(world))) < 1e-6     # the non-decoder remainder
    # gradient accumulation: the first micro-batch outside the reducer (local accumulation, like DDP.no_sync), the
    # second inside -- the decoders' .grad already exist then, so their accumulated TOTALS are reduced with the remainder:
    # twice the single-step result, nothing overwritten
    for q in holder.parameters():
        q.grad = None

    def micro():
        zo, ldo = ms.forward_all(ptm, gtm, 'inverse')
        extra = holder[-1](torch.ones(1, 3, device='cuda')).sum() * (rank + 1.0)
        ((zo * wk * torch.from_numpy(wz[b0:b1]).cuda()).sum() + (ldo * wk * torch.from_numpy(wl[b0:b1]).cuda()).sum() + extra).backward()
    micro()
    launched_before = reducer.launched
    with reducer:
        micro()
    assert reducer.launched == launched_before          # accumulating: no decoder was reduced from inside the backward pass
    got2 = torch.cat([q.grad.reshape(-1) for d in decs for q in d.parameters()])
    assert float((got2 - 2.0 * ref_grads).abs().max() / ref_grads.abs().max()) < 2e-5
    assert abs(float(holder[-1].bias.grad[0]) - 2.0 * sum(r + 1.0 for r in range(world))) < 1e-5
    for d in decs:                                       # the saved comparison below reads the single-step gradients
        for q in d.parameters():
            q.grad = None
    with reducer:
        micro()
    assert reducer.launched == launched_before + 2
    mix = {'zm': zm.detach().cpu().numpy(), 'dpm': dpm, 'dgm': dgm,
           'gm': torch.cat([q.grad.reshape(-1) for d in decs for q in d.parameters()]).cpu().numpy()}
    n_flat = all_reduce_gradients(m, average=False)           # one flat buffer, summed (DDP would also divide by W)
    assert n_flat == sum(q.numel() for q in m.parameters())
    grads = torch.cat([q.grad.reshape(-1) for q in m.parameters()])
    out = {'z': z.detach().cpu().numpy(), 'dp': pt.grad.cpu().numpy(), 'dg': gt.grad.cpu().numpy(),
           'rv': torch.cat([v.reshape(-1) for k, v in m.state_dict().items() if k.endswith('running_var')]).cpu().numpy()}
    np.savez(os.path.join(os.environ['GWTF_TMP'], f'rank{rank}.npz'), grads=grads.cpu().numpy(), b0=b0, b1=b1, **out, **mix)
    # the reference's own wrapping (train_ae.py:152-153): SyncBatchNorm conversion + DistributedDataParallel
    m2, _ = decoder_and_state(L, f, G, 13)
    m2 = torch.nn.SyncBatchNorm.convert_sync_batchnorm(m2.cuda().train())
    ddp = torch.nn.parallel.DistributedDataParallel(m2, find_unused_parameters=True)
    pt2 = torch.from_numpy(p[b0:b1]).cuda().requires_grad_(True)
    gt2 = torch.from_numpy(g[b0:b1]).cuda().requires_grad_(True)
    ps2, mus2, lvs2 = ddp(pt2, gt2, mode='inverse')
    loss2 = (ps2[0] * torch.from_numpy(wz[b0:b1]).cuda()).sum() + (sum(lvs2) * torch.from_numpy(wl[b0:b1]).cuda()).sum()
    loss2.backward()
    ddp_grads = torch.cat([q.grad.reshape(-1) for q in m2.parameters()]) * world      # DDP averages; undo for comparison
    ddp_err = float((ddp_grads - grads).abs().max() / grads.abs().max())
    assert ddp_err < 1e-5, ddp_err
    dist.barrier()
    if rank == 0:
        print(f'DDP wrapped: parameter gradients == flat all-reduce path (rel {ddp_err:.1e})', flush=True)
        dist.destroy_process_group()
        m1, _ = decoder_and_state(L, f, G, 13)
        m1 = m1.cuda().train()
        pf, gf = torch.from_numpy(p).cuda().requires_grad_(True), torch.from_numpy(g).cuda().requires_grad_(True)
        z1, ld1 = m1.forward_fused(pf, gf, 'inverse')
        ((z1 * torch.from_numpy(wz).cuda()).sum() + (ld1 * torch.from_numpy(wl).cuda()).sum()).backward()
        g1 = torch.cat([q.grad.reshape(-1) for q in m1.parameters()]).cpu().numpy()
        parts = [np.load(os.path.join(os.environ['GWTF_TMP'], f'rank{r}.npz')) for r in range(world)]
        rel = lambda a, b: float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))
        res = {
            'z': rel(np.concatenate([q['z'] for q in parts]), z1.detach().cpu().numpy()),
            'dp': rel(np.concatenate([q['dp'] for q in parts]), pf.grad.cpu().numpy()),
            'dg': rel(np.concatenate([q['dg'] for q in parts]), gf.grad.cpu().numpy()),
            'param_grads': rel(parts[0]['grads'], g1),
            'running_var': rel(parts[0]['rv'], torch.cat([v.reshape(-1) for k, v in m1.state_dict().items()
                                                          if k.endswith('running_var')]).cpu().numpy()),
        }
        # the K = 2 mixture, single process
        decs1 = [decoder_and_state(L, f, G, 20 + k)[0].cuda().train() for k in range(2)]
        ms1 = gw.MixtureStack(decs1)
        pm, gm_ = torch.from_numpy(p).cuda().requires_grad_(True), torch.from_numpy(g).cuda().requires_grad_(True)
        zm1, ldm1 = ms1.forward_all(pm, gm_, 'inverse')
        ((zm1 * wk * torch.from_numpy(wz).cuda()).sum() + (ldm1 * wk * torch.from_numpy(wl).cuda()).sum()).backward()
        res['mix_z'] = rel(np.concatenate([q['zm'] for q in parts], axis=1), zm1.detach().cpu().numpy())
        res['mix_dp'] = rel(np.concatenate([q['dpm'] for q in parts]), pm.grad.cpu().numpy())
        res['mix_dg'] = rel(np.concatenate([q['dgm'] for q in parts]), gm_.grad.cpu().numpy())
        res['mix_param_grads'] = rel(parts[0]['gm'], torch.cat([q.grad.reshape(-1) for d in decs1 for q in d.parameters()]).cpu().numpy())
        print('DIST2', ' '.join(f'{k}={v:.2e}' for k, v in res.items()), flush=True)
        assert all(v < 2e-3 for v in res.values()), res
    else:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
