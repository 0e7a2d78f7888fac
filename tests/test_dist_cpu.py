"""N>1 path on CPU: two gloo processes shard the batch of shapes exactly like the reference and agree with the
single-process result (forward needs no exchange); the bench timing reduction takes the max over ranks."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from go_with_the_flows_amd.dist import shard_bounds


def test_shard_bounds_partition():
    for B in (1, 7, 32, 64, 128, 130):
        for W in (1, 2, 3, 8):
            cuts = [shard_bounds(B, r, W) for r in range(W)]
            assert cuts[0][0] == 0 and cuts[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            sizes = [e - b for b, e in cuts]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)   # +1 on the low ranks


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from go_with_the_flows_amd.dist import init_from_env, max_over_ranks, shard_bounds, sum_over_ranks
    from go_with_the_flows_amd.synth import synth_state, synth_inputs
    import go_with_the_flows_amd as gw
    from oracle import flow_oracle as fo
    r, w = init_from_env('gloo')
    assert (r, w) == (rank, world)
    L, f, G, B, N = 1, 8, 16, 5, 24
    st = synth_state(gw.LocalCondRNVPDecoder(L, f, G).state_dict(), 3)
    p, g = synth_inputs(B, N, G, 4)
    b0, b1 = shard_bounds(B, rank, world)
    out, ld = fo.decoder_fused(p[b0:b1], g[b0:b1], st, L, 'inverse')      # this rank's shard (oracle stands in for the GPU)
    np.savez(os.path.join(tmp, f'r{rank}.npz'), out=out, ld=ld, b0=b0, b1=b1)
    # per-rank log-likelihood partial sums reduce to the global sum
    s = sum_over_ranks(torch.tensor([float(ld.sum())], dtype=torch.float64))
    t = max_over_ranks(1.0 + rank)
    dist.barrier()
    if rank == 0:
        full_out, full_ld = fo.decoder_fused(p, g, st, L, 'inverse')
        parts = [np.load(os.path.join(tmp, f'r{k}.npz')) for k in range(world)]
        assert np.array_equal(np.concatenate([q['out'] for q in parts]), full_out)
        assert np.array_equal(np.concatenate([q['ld'] for q in parts]), full_ld)
        assert abs(float(s) - float(full_ld.astype(np.float64).sum())) < 1e-4
        assert t == float(world)
    dist.destroy_process_group()


def test_two_rank_gloo_sharding(tmp_path):
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)


def _direct_worker(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from go_with_the_flows_amd.dist import init_from_env, all_reduce_direct, all_reduce_gradients
    init_from_env('gloo')
    for n in (1, 2, 5, 1000, 4099):                      # fewer elements than ranks, ragged last chunk, empty chunks
        g = torch.Generator().manual_seed(100 * n + rank)
        mine = torch.randn(n, generator=g)
        want = mine.clone()
        dist.all_reduce(want, op=dist.ReduceOp.SUM)
        got = all_reduce_direct(mine.clone())
        assert torch.allclose(got, want, rtol=1e-6, atol=1e-6), (n, rank)
        # every rank must end with bit-identical values (DDP semantics: replicas stay in sync)
        ref = got.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(got, ref), (n, rank)
    # through the module-level entry point: gradients averaged over the ranks
    lin = torch.nn.Linear(7, 3)
    with torch.no_grad():
        for p in lin.parameters():
            p.grad = torch.full_like(p, float(rank + 1))
    n_flat = all_reduce_gradients(lin, average=True, algorithm='direct')
    assert n_flat == 7 * 3 + 3
    mean = sum(r + 1.0 for r in range(world)) / world
    assert all(torch.allclose(p.grad, torch.full_like(p, mean)) for p in lin.parameters())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_direct_reduce_scatter_all_gather_equals_all_reduce(world):
    """The point-to-point (xGMI-mesh) gradient exchange against the library all-reduce, 2 and 3 gloo ranks."""
    port = 31500 + os.getpid() % 2000 + world
    mp.spawn(_direct_worker, args=(world, port), nprocs=world, join=True)
