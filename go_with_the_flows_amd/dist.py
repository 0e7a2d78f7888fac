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


def all_reduce_gradients(module, average=True, force=False):
    """Data-parallel gradient exchange as ONE flat fp32 buffer (19-104 MB per step for the shipped
    configs): flatten every .grad, a single all-reduce over the default group (RCCL on ROCm: backend 'nccl'), divide
    by the world size (DDP semantics, reference train_ae.py:153), scatter back.  A drop-in for DistributedDataParallel
    when the model is used without the DDP wrapper; with DDP, its bucketed all-reduce does the same job."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return 0          # force=True still runs the collective on a 1-rank group (exercises the RCCL path on one GPU)
    params = [p for p in module.parameters() if p.grad is not None]
    if not params:
        return 0
    grads = [p.grad for p in params]
    flat = torch.cat([g.reshape(-1) for g in grads])
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

