"""One-process-per-GPU plumbing: batch-of-shapes sharding and the barrier/max timing used by bench.py.

The path needs no data exchange in the forward direction (every shape is independent given g): ranks own
disjoint slices of the batch, exactly as the reference shards it (DistributedSampler + per-rank batch
``B // world`` with the remainder on the low ranks, train_ae.py:77-78,100-109).  Backend 'nccl' is RCCL on
ROCm; 'gloo' is used by the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def shard_bounds(n_shapes, rank, world):
    """[begin, end) of the shapes rank owns: B//world each, +1 for the first B % world ranks (train_ae.py:77-78)."""
    base, rem = divmod(n_shapes, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def init_from_env(backend=None, device=None):
    """init_process_group from torchrun's env (RANK/WORLD_SIZE/MASTER_*); returns (rank, world)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = backend or ('nccl' if torch.cuda.is_available() else 'gloo')
        kw = {'device_id': device} if (backend == 'nccl' and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return rank, world


def max_over_ranks(value, device='cpu'):
    """MAX-all-reduce of a python float (bench timing contract)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(t):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def all_reduce_direct(flat, group=None):
    """Sum `flat` over the ranks with the DIRECT algorithm: reduce-scatter + all-gather done as point-to-point
    messages between every pair of ranks, not around a ring.  On an MI355X node every GPU has its own xGMI link to each of the
    other seven (7 x ~153 GB/s, no switch), so a rank can send its W - 1 chunks to their owners at the same time over W - 1
    different links and the whole exchange is two steps of n / W elements per link -- a ring all-reduce moves 2 (W - 1) / W n
    elements through EVERY link in 2 (W - 1) dependent steps.  In place; `flat` is a contiguous 1-D tensor of any length (the
    chunk grid is padded on the fly).  Uses only batched isend / irecv, which RCCL ('nccl') and gloo both provide."""
    world = dist.get_world_size(group)
    if world == 1:
        return flat
    rank = dist.get_rank(group)
    n = flat.numel()
    chunk = (n + world - 1) // world
    bounds = [(min(n, r * chunk), min(n, (r + 1) * chunk)) for r in range(world)]
    lo, hi = bounds[rank]
    peers = [r for r in range(world) if r != rank]
    to_global = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    # step 1 (reduce-scatter): chunk r of every rank goes to rank r; mine arrives from everybody else
    inbox = [torch.empty(hi - lo, dtype=flat.dtype, device=flat.device) for _ in peers]
    ops = []
    for r, buf in zip(peers, inbox):
        a, b = bounds[r]
        if b > a:
            ops.append(dist.P2POp(dist.isend, flat[a:b], to_global(r), group))
        if hi > lo:
            ops.append(dist.P2POp(dist.irecv, buf, to_global(r), group))
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()
    mine = flat[lo:hi]
    for buf in inbox:                      # fixed order of additions: rank 0's, rank 1's, ... -- the same sum on every run
        mine.add_(buf)
    # step 2 (all-gather): the reduced chunk goes to everybody, theirs arrive in place
    ops = []
    for r in peers:
        a, b = bounds[r]
        if hi > lo:
            ops.append(dist.P2POp(dist.isend, mine, to_global(r), group))
        if b > a:
            ops.append(dist.P2POp(dist.irecv, flat[a:b], to_global(r), group))
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()
    return flat


def all_reduce_gradients(module, average=True, force=False, algorithm='ring'):
    """Data-parallel gradient exchange as ONE flat fp32 buffer (19-104 MB per step for the shipped
    configs): flatten every .grad, a single all-reduce over the default group (RCCL on ROCm: backend 'nccl'), divide
    by the world size (DDP semantics, reference train_ae.py:153), scatter back.  A drop-in for DistributedDataParallel
    when the model is used without the DDP wrapper; with DDP, its bucketed all-reduce does the same job.
    algorithm: 'ring' = the library's all-reduce (RCCL picks its own schedule); 'direct' = all_reduce_direct, the point-to-point
    reduce-scatter + all-gather sized for the xGMI mesh."""
    if algorithm not in ('ring', 'direct'):
        raise ValueError(f'unknown gradient exchange algorithm {algorithm!r}')
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return 0          # force=True still runs the collective on a 1-rank group (exercises the RCCL path on one GPU)
    params = [p for p in module.parameters() if p.grad is not None]
    if not params:
        return 0
    grads = [p.grad for p in params]
    flat = torch.cat([g.reshape(-1) for g in grads])
    if algorithm == 'direct':
        all_reduce_direct(flat)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat.div_(dist.get_world_size())
    # scatter back with ONE multi-tensor copy (a per-parameter copy_ is ~650 launches for the airplane model)
    torch._foreach_copy_(grads, [v.view_as(g) for v, g in zip(flat.split_with_sizes([g.numel() for g in grads]), grads)])
    return flat.numel()


class OverlappedGradients:
    """Data-parallel gradient exchange overlapped with the backward pass (the job DistributedDataParallel's buckets do in
    the reference, train_ae.py:153).  The decoders hold ~3/4 of the model's parameters (3.7 M of 4.8 M for the airplane
    config) and are the LAST modules of the forward pass, so their gradients are complete FIRST in the backward pass -- one
    flat tensor per decoder (the gradient of its raw parameter arena).  Their all-reduces are launched asynchronously
    right there (RCCL runs it on its own stream over xGMI) while the encoder / prior-flow backward continues on the compute
    stream; ``finish()`` reduces what is left as one more flat buffer and waits.

        reducer = OverlappedGradients(model)          # once
        with reducer:                                  # per step
            loss.backward()
        optimizer.step()
    """

    def __init__(self, module, average=True):
        self.module, self.average = module, average
        self.pending = []
        self.launched = 0          # asynchronous collectives launched from inside the backward pass (tests read it)

    def __enter__(self):
        from . import autograd
        self.pending = []
        autograd.GRAD_SINK['reducer'] = self if (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) else None
        return self

    def __exit__(self, exc_type, exc, tb):
        from . import autograd
        autograd.GRAD_SINK['reducer'] = None
        if exc_type is None:
            self.finish()
        return False

    def on_flat_gradient(self, grad, engine):
        grad = grad.contiguous()
        work = dist.all_reduce(grad, op=dist.ReduceOp.SUM, async_op=True)     # in place, on the collective stream
        self.pending.append((work, grad, engine))
        self.launched += 1
        return grad

    def finish(self):
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        covered, dsts, srcs = set(), [], []
        for work, grad, e in self.pending:
            work.wait()
            if self.average:
                grad.div_(world)
            for (t, _op), v in zip(e._srcs, grad.split_with_sizes(e._sizes)):
                if t is None or not t.requires_grad or t.grad is None:
                    continue
                covered.add(id(t))
                if t.grad.data_ptr() != v.data_ptr():      # autograd cloned instead of keeping the view: copy the reduced values
                    dsts.append(t.grad)
                    srcs.append(v.view_as(t.grad))
        if dsts:
            torch._foreach_copy_(dsts, srcs)
        self.pending = []
        if world == 1:
            return
        rest = [p.grad for p in self.module.parameters() if p.grad is not None and id(p) not in covered]
        if rest:
            flat = torch.cat([g.reshape(-1) for g in rest])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            if self.average:
                flat.div_(world)
            torch._foreach_copy_(rest, [v.view_as(g) for v, g in zip(flat.split_with_sizes([g.numel() for g in rest]), rest)])

