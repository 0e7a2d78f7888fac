"""Worker for tests/test_gpu_encoder.py::test_two_rank_syncbn_encoder_matches_single_process.
Two ranks share cuda:0 (one-GPU box), gloo carries the statistic sums.  The encoder is converted to SyncBatchNorm as the
reference does (train_ae.py:152); each rank owns half of the batch; pooled codes, running statistics and the parameter
gradients summed over the ranks must equal the single-process full-batch run of the HIP train pipeline."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from go_with_the_flows_amd import encoders                   # noqa: E402
from go_with_the_flows_amd.dist import shard_bounds          # noqa: E402
from go_with_the_flows_amd.synth import load_synth_, synth_inputs  # noqa: E402


def build():
    m = encoders.PointNetCloudEncoder(3, 64, [128, 256, 512])
    load_synth_(m, 41)
    return m.cuda().train()


def main():
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    B, N = 6, 136
    x = synth_inputs(B, N, 4, 42)[0]
    wgt = np.random.default_rng(43).standard_normal((B, 512)).astype(np.float32)
    b0, b1 = shard_bounds(B, rank, world)
    m = torch.nn.SyncBatchNorm.convert_sync_batchnorm(build())
    xs = torch.from_numpy(x[b0:b1]).cuda()
    assert m._train_pipeline_ok(xs) and encoders._bn_sync(m.features.init_sd_bn)
    pooled = m.forward_max(xs)
    assert 'EncoderTrainFn' in type(pooled.grad_fn).__name__
    (pooled * torch.from_numpy(wgt[b0:b1]).cuda()).sum().backward()
    grads = torch.cat([q.grad.reshape(-1) for q in m.parameters()])
    from go_with_the_flows_amd.dist import all_reduce_direct
    direct = all_reduce_direct(grads.clone())                  # point-to-point reduce-scatter + all-gather on device tensors
    dist.all_reduce(grads)                                     # what DDP does (it also divides by the world size)
    assert float((direct - grads).abs().max()) <= 1e-6 * float(grads.abs().max())
    bufs = torch.cat([v.reshape(-1).float() for k, v in m.state_dict().items() if 'running' in k])
    np.savez(os.path.join(os.environ['GWTF_TMP'], f'enc{rank}.npz'), pooled=pooled.detach().cpu().numpy(),
             grads=grads.cpu().numpy(), bufs=bufs.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        m1 = build()
        x1 = torch.from_numpy(x).cuda()
        p1 = m1.forward_max(x1)
        (p1 * torch.from_numpy(wgt).cuda()).sum().backward()
        g1 = torch.cat([q.grad.reshape(-1) for q in m1.parameters()]).cpu().numpy()
        bf1 = torch.cat([v.reshape(-1).float() for k, v in m1.state_dict().items() if 'running' in k]).cpu().numpy()
        parts = [np.load(os.path.join(os.environ['GWTF_TMP'], f'enc{r}.npz')) for r in range(world)]
        rel = lambda a, b: float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))
        res = {'pooled': rel(np.concatenate([q['pooled'] for q in parts]), p1.detach().cpu().numpy()),
               'grads': rel(parts[0]['grads'], g1), 'running': rel(parts[0]['bufs'], bf1)}
        print('ENC2', ' '.join(f'{k}={v:.2e}' for k, v in res.items()), flush=True)
        assert all(v < 1e-4 for v in res.values()), res


if __name__ == '__main__':
    main()
