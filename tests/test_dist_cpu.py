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
    # the overlapped reducer with the exchange algorithm as a constructor argument: same averaged gradients
    from go_with_the_flows_amd.dist import OverlappedGradients
    for algo in ('ring', 'direct'):
        torch.manual_seed(5)
        net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Tanh(), torch.nn.Linear(5, 2))
        red = OverlappedGradients(net, algorithm=algo)
        assert red.algorithm == algo
        x = torch.randn(4, 7, generator=torch.Generator().manual_seed(50 + rank))
        with red:
            net(x).square().sum().backward()
        got = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
        net.zero_grad(set_to_none=True)
        tot = None
        for r in range(world):                                   # every rank's local gradient, recomputed here
            xr = torch.randn(4, 7, generator=torch.Generator().manual_seed(50 + r))
            net(xr).square().sum().backward()
        want = torch.cat([p.grad.reshape(-1) for p in net.parameters()]) / world
        assert torch.allclose(got, want, rtol=1e-5, atol=1e-6), (algo, rank)
    with pytest.raises(ValueError):
        OverlappedGradients(lin, algorithm='tree', sync_state=False)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_direct_reduce_scatter_all_gather_equals_all_reduce(world):
    """The point-to-point (xGMI-mesh) gradient exchange against the library all-reduce, 2 and 3 gloo ranks."""
    port = 31500 + os.getpid() % 2000 + world
    mp.spawn(_direct_worker, args=(world, port), nprocs=world, join=True)


def _rows_worker(rank, world, port):
    """dist.row_layout / gather_rows (the rows of ALL ranks for the per-shape modules) and the FeatureEncoder head on top of them:
    ragged per-rank batches, forward == the single-process full batch, backward == its gradients (SyncBatchNorm semantics)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from go_with_the_flows_amd import dist as gd
    gd.init_from_env('gloo')
    B, G = 7, 6
    b0, b1 = gd.shard_bounds(B, rank, world)
    full = torch.randn(B, G, generator=torch.Generator().manual_seed(5))
    lay = gd.row_layout(b1 - b0, torch.device('cpu'))
    assert lay.sizes == [gd.shard_bounds(B, r, world)[1] - gd.shard_bounds(B, r, world)[0] for r in range(world)]
    assert (lay.row0, lay.total, lay.even) == (b0, B, B % world == 0)
    assert gd.row_layout(b1 - b0, torch.device('cpu')) is lay                    # unchanged sizes: the same layout object
    mine = full[b0:b1].clone().requires_grad_(True)
    rows, lay2 = gd.gather_rows(mine)
    assert torch.equal(rows, full) and lay2 is lay
    w = torch.arange(B * G, dtype=torch.float32).view(B, G) * (rank + 1)          # every rank weighs every row differently
    (rows * w).sum().backward()
    want = sum(torch.arange(B * G, dtype=torch.float32).view(B, G) * (r + 1) for r in range(world))[b0:b1]
    assert torch.allclose(mine.grad, want)
    # a per-shape head converted to SyncBatchNorm (train_ae.py:152): every rank runs the trunk on all rows and keeps its own
    from go_with_the_flows_amd.encoders import FeatureEncoder
    torch.manual_seed(3)
    head = FeatureEncoder(2, G, 3)
    ref = FeatureEncoder(2, G, 3)
    ref.load_state_dict(head.state_dict())
    head = torch.nn.SyncBatchNorm.convert_sync_batchnorm(head).train()
    x = full[b0:b1].clone().requires_grad_(True)
    mu, lv = head(x, bn_updates=2)
    wr = torch.linspace(0.5, 1.5, B).view(B, 1)
    ((mu + lv * lv) * wr[b0:b1]).sum().backward()
    for q in head.parameters():
        dist.all_reduce(q.grad)
    xf = full.clone().requires_grad_(True)
    ref.train()
    mu1, lv1 = ref(xf, bn_updates=2)
    ((mu1 + lv1 * lv1) * wr).sum().backward()
    assert torch.allclose(mu, mu1[b0:b1], atol=1e-6) and torch.allclose(lv, lv1[b0:b1], atol=1e-6)
    assert torch.allclose(x.grad, xf.grad[b0:b1], atol=1e-5)
    for q, q1 in zip(head.parameters(), ref.parameters()):
        assert torch.allclose(q.grad, q1.grad, atol=1e-5)
    for (k, v), v1 in zip(head.state_dict().items(), ref.state_dict().values()):
        assert torch.allclose(v.float(), v1.float(), atol=1e-6), k                 # running statistics, num_batches_tracked (+2)
    # ... and the two-update replay equals two real passes
    two = FeatureEncoder(2, G, 3)
    two.load_state_dict({k: v for k, v in zip(two.state_dict(), [t.clone() for t in FeatureEncoder(2, G, 3).state_dict().values()])})
    one = FeatureEncoder(2, G, 3)
    one.load_state_dict(two.state_dict())
    two.train(), one.train()
    two(full), two(full)
    one(full, bn_updates=2)
    for (k, v), v1 in zip(two.state_dict().items(), one.state_dict().values()):
        assert torch.allclose(v.float(), v1.float(), atol=1e-6), k
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_row_layout_and_gather_rows(world):
    port = 33500 + os.getpid() % 2000 + world
    mp.spawn(_rows_worker, args=(world, port), nprocs=world, join=True)


def _state_worker(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from go_with_the_flows_amd.dist import init_from_env, sync_module_state, OverlappedGradients
    init_from_env('gloo')
    torch.manual_seed(100 + rank)                       # the reference sets no seed: every rank builds a DIFFERENT model
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.BatchNorm1d(7), torch.nn.Linear(7, 3))
    net[1].running_mean.normal_()
    net[1].num_batches_tracked += 3 + rank
    mine = torch.cat([p.detach().reshape(-1) for p in net.parameters()] + [net[1].running_mean])
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    assert not torch.equal(gathered[0], gathered[1])    # they do start apart
    try:
        sync_module_state(net, verify_only=True)
        raise AssertionError('a model mismatch between the ranks must be reported')
    except RuntimeError as e:
        assert 'differs between the ranks' in str(e)
    red = OverlappedGradients(net)                      # the data-parallel wrapper synchronises at construction, like DDP
    assert red.synced == sum(p.numel() for p in net.parameters()) + sum(b.numel() for b in net.buffers())
    mine = torch.cat([p.detach().reshape(-1) for p in net.parameters()] + [net[1].running_mean, net[1].running_var,
                                                                            net[1].num_batches_tracked.float().reshape(1)])
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    assert all(torch.equal(gathered[0], q) for q in gathered[1:])
    assert int(net[1].num_batches_tracked) == 3          # rank 0's
    sync_module_state(net, verify_only=True)            # and the check is silent now
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_data_parallel_wrapper_starts_every_rank_from_rank0_state(world):
    """DistributedDataParallel broadcasts rank 0's parameters and buffers at construction (train_ae.py:153); the
    reference sets no seed, so the hand-rolled reducer must do the same or ranks train different models with averaged gradients."""
    port = 31500 + os.getpid() % 2000 + world
    mp.spawn(_state_worker, args=(world, port), nprocs=world, join=True)


def _row_layout_worker(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from go_with_the_flows_amd import dist as gd
    gd.init_from_env('gloo')
    cpu = torch.device('cpu')
    lay = gd.row_layout(5, cpu)                       # both ranks hold 5 rows
    assert lay.sizes == [5, 5] and lay.total == 10 and lay.row0 == 5 * rank
    n0 = gd.ROW_EXCHANGES['n']
    for _ in range(3):
        assert gd.row_layout(5, cpu) is lay           # unchanged sizes: the same object ...
    assert gd.ROW_EXCHANGES['n'] == n0 + 3            # ... but every eager lookup exchanged the sizes again
    # rank 1's loader now delivers 3 rows while rank 0 keeps 5.  Rank 0's own count is unchanged -- with a cache keyed on it (rounds
    # 3-4) rank 0 went straight to the row gather while rank 1 exchanged sizes: mismatched collectives.  Every lookup exchanges
    # now, so both ranks learn the new layout in the SAME collective, whichever rank's batch changed.
    now = gd.row_layout(3 if rank == 1 else 5, cpu)
    assert now.sizes == [5, 3] and now.total == 8 and now.row0 == (0 if rank == 0 else 5)
    rows, lay2 = gd.gather_rows(torch.full((3 if rank == 1 else 5, 2), float(rank)))
    assert lay2.sizes == [5, 3] and torch.equal(rows, torch.cat([torch.zeros(5, 2), torch.ones(3, 2)]))
    # back to 5 / 5: the remembered layout of this rank's 5 rows had become [5, 3] on rank 0 -- it is replaced, not trusted
    assert gd.row_layout(5, cpu).sizes == [5, 5]
    # opt-out for fixed-batch eager loops: exchange once per batch size, cached afterwards (no collective, no protection)
    os.environ['GWTF_ROW_LAYOUT_TRUST_CACHE'] = '1'
    n1 = gd.ROW_EXCHANGES['n']
    assert gd.row_layout(5, cpu).sizes == [5, 5] and gd.ROW_EXCHANGES['n'] == n1
    del os.environ['GWTF_ROW_LAYOUT_TRUST_CACHE']
    dist.barrier()
    dist.destroy_process_group()


def test_row_layout_cache_notices_a_batch_size_change_on_another_rank():
    """the size exchange is unconditional outside graph captures, so a batch size that changes on ONE rank reaches every
    rank in the same collective (the periodic re-check of round 4 only worked when the change landed on a rank's 64th lookup)."""
    port = 29500 + (os.getpid() + 977) % 2000
    mp.spawn(_row_layout_worker, args=(2, port), nprocs=2, join=True)
