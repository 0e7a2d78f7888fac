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


def capture_mode():
    """capture_error_mode for torch.cuda.graph(...).  With a process group alive, ProcessGroupNCCL's watchdog THREAD polls the
    events of earlier collectives (hipEventQuery); under the default 'global' mode that call is illegal while this thread
    captures and takes the process down ("operation not permitted when stream is capturing").  'thread_local' restricts the
    check to the capturing thread."""
    return 'thread_local' if (dist.is_available() and dist.is_initialized()) else 'global'


# ---- the package's own eager collectives are issued through `run` -------------------------------------------------------------
# Eager (not captured) RCCL collectives are issued asynchronously and waited for on the current stream at once -- the same ordering
# as the blocking form -- so that the package HOLDS their Work objects: quiesce_for_capture polls exactly those instead of sleeping
# for a guessed number of watchdog periods.  Inside a capture, and on gloo, the blocking form is used (a Work created inside a
# capture would sit in ProcessGroupNCCL's watchdog list with events that cannot be queried).
_EAGER_WORKS = []


def _capturing(t=None):
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def run(fn, *args, **kwargs):
    """fn = a torch.distributed collective (all_reduce, all_gather, all_gather_into_tensor, all_to_all_single, broadcast)."""
    if dist.get_backend() != 'nccl' or _capturing():
        return fn(*args, **kwargs)
    w = fn(*args, async_op=True, **kwargs)
    if w is not None:
        w.wait()                                    # stream-ordered (the host does not block): what the blocking form does
        if len(_EAGER_WORKS) >= 64:                 # keep the list short: finished ones go
            _EAGER_WORKS[:] = [x for x in _EAGER_WORKS if not x.is_completed()]
        _EAGER_WORKS.append(w)
    return None


def quiesce_for_capture(device=None, timeout_s=10.0):
    """Call right before a hipGraph capture in a process with an RCCL process group.  ProcessGroupNCCL's watchdog thread polls the
    completion events of earlier (eager) collectives; a capture must not begin while one of them is still in flight.  Drain: finish
    all device work, then poll is_completed() of every Work this package issued (`run` above) until all report done -- a loop on the
    objects themselves, not a sleep for a guessed number of watchdog periods (rounds 3-4 slept 0.35 s).  GWTF_CAPTURE_QUIESCE_S adds
    a fixed grace period on top (default 0) for torch builds whose watchdog needs one.  The primary protection stays
    capture_mode() = 'thread_local'."""
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == 'nccl' and torch.cuda.is_available():
        import time
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        while any(not w.is_completed() for w in _EAGER_WORKS):
            if time.perf_counter() - t0 > timeout_s:
                raise RuntimeError(f'{sum(not w.is_completed() for w in _EAGER_WORKS)} eager collective(s) still pending after {timeout_s} s: '
                                   'a rank is missing from a collective; a hipGraph capture cannot start')
            time.sleep(0.002)
        del _EAGER_WORKS[:]
        grace = float(os.environ.get('GWTF_CAPTURE_QUIESCE_S', '0'))
        if grace > 0:
            time.sleep(grace)


def graph_capture(graph, device=None, **kwargs):
    """torch.cuda.graph(graph) made safe beside a live process group: quiesce_for_capture + capture_mode()."""
    quiesce_for_capture(device)
    return torch.cuda.graph(graph, capture_error_mode=capture_mode(), **kwargs)


def sum_over_ranks(t):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        run(dist.all_reduce, t, op=dist.ReduceOp.SUM)
    return t


def sharded():
    """True when batch statistics must be summed over the ranks of torch.distributed's default group: more than one rank, or
    GWTF_FORCE_SHARDED=1 (takes the multi-rank code path on a 1-rank group -- the only way to run the collectives over RCCL,
    and to capture them in a hipGraph, on a one-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get('GWTF_FORCE_SHARDED') == '1'


def syncs_statistics(bn_modules):
    """True when these BatchNorm modules must see the statistics of ALL ranks: the reference converts every BatchNorm to
    SyncBatchNorm before wrapping the model in DistributedDataParallel (train_ae.py:152), and the run is sharded()."""
    return sharded() and any(isinstance(m, torch.nn.SyncBatchNorm) for m in bn_modules)


# ---- the rows (shapes) each rank holds -------------------------------------------------------------------------------------
# Per-shape work (the FiLM heads of the decoders, the prior flow on the latent, the Gaussian heads) is a few MFLOP on B rows:
# every rank runs it on the rows of ALL ranks (one all-gather of the B x G latents) instead of synchronising the statistics of
# each of its BatchNorm layers -- same numbers as SyncBatchNorm, two collectives per module instead of two per layer.  That
# needs every rank's row count.  Eager steps exchange the counts at EVERY lookup (one 8-byte all-gather: every rank enters the
# same collective whatever its own batch did, so a batch size that changes on one rank only can never pair a size exchange on
# that rank with a row gather on another); a captured hipGraph step cannot talk to the host and replays the layout of the eager
# warm-up step that preceded its capture -- its loader must keep the per-rank batch fixed, as the reference's do (DistributedSampler
# + drop_last, train_ae.py:77-78,97-109).
class RowLayout:
    __slots__ = ('sizes', 'row0', 'total', 'even')

    def __init__(self, sizes, rank):
        self.sizes = [int(s) for s in sizes]
        self.row0, self.total = sum(self.sizes[:rank]), sum(self.sizes)
        self.even = len(set(self.sizes)) == 1

    def __repr__(self):
        return f'RowLayout(sizes={self.sizes}, row0={self.row0})'


_ROW_LAYOUTS = {}
ROW_EXCHANGES = {'n': 0}      # size exchanges done by this process (tests read it)


def reset_row_layouts():
    """Forget the cached per-rank batch sizes (they are only ever used inside hipGraph captures and under
    GWTF_ROW_LAYOUT_TRUST_CACHE=1; kept for callers of earlier versions)."""
    _ROW_LAYOUTS.clear()


def _exchange_sizes(rows, device, world, rank):
    on = device if dist.get_backend() == 'nccl' else torch.device('cpu')
    sizes = [torch.zeros(1, dtype=torch.int64, device=on) for _ in range(world)]
    run(dist.all_gather, sizes, torch.tensor([int(rows)], dtype=torch.int64, device=on))
    ROW_EXCHANGES['n'] += 1
    return RowLayout([int(x.item()) for x in sizes], rank)


def row_layout(rows, device):
    """The number of rows every rank of the default group holds, given that this rank holds `rows`.
    Outside hipGraph captures this is a collective on EVERY call (all ranks must call it in the same order -- they do: the calls
    are the model's per-step module calls) and its result is remembered under this rank's row count.  Inside a capture nothing
    can be exchanged: the remembered layout of the last eager call with the same `rows` is used (GraphedTrainStep and every
    capture of this package run an eager warm-up step first) and a missing one raises.
    GWTF_ROW_LAYOUT_TRUST_CACHE=1 (set on ALL ranks or on none) restores exchange-once-per-batch-size for eager loops whose
    per-rank batch never changes: no device-to-host synchronisation after the first step, and no protection either."""
    if not (dist.is_available() and dist.is_initialized()):
        return RowLayout([rows], 0)
    world, rank = dist.get_world_size(), dist.get_rank()
    if world == 1:
        return RowLayout([rows], 0)
    key = (world, rank, int(rows))
    capturing = device.type == 'cuda' and torch.cuda.is_current_stream_capturing()
    lay = _ROW_LAYOUTS.get(key)
    if capturing:
        if lay is None:
            raise RuntimeError('the per-rank batch sizes are exchanged by eager steps: run one step with this batch size outside '
                               'the hipGraph capture first (GraphedTrainStep does)')
        return lay
    if lay is not None and os.environ.get('GWTF_ROW_LAYOUT_TRUST_CACHE') == '1':
        return lay
    now = _exchange_sizes(rows, device, world, rank)
    if lay is not None and lay.sizes == now.sizes:
        return lay                                          # unchanged: callers may compare layouts by identity
    _ROW_LAYOUTS[key] = now
    return now


def _gather_padded(t, lay):
    """all_gather of (B_r, ...) blocks whose B_r may differ (collectives need equal shapes: pad to the largest, then trim)."""
    bmax = max(lay.sizes)
    mine = t.contiguous()
    if mine.shape[0] < bmax:
        mine = torch.cat([mine, mine.new_zeros((bmax - mine.shape[0],) + tuple(mine.shape[1:]))])
    out = mine.new_empty((len(lay.sizes) * bmax,) + tuple(mine.shape[1:]))
    if dist.get_backend() == 'nccl':
        run(dist.all_gather_into_tensor, out, mine)
    else:
        dist.all_gather(list(out.chunk(len(lay.sizes))), mine)
    if lay.even:
        return out
    return torch.cat([blk[:n] for blk, n in zip(out.chunk(len(lay.sizes)), lay.sizes)])


class GatherRows(torch.autograd.Function):
    """rows of all ranks, concatenated in rank order: (B_r, ...) -> (sum_r B_r, ...).  The backward sums the gradient every
    rank holds for every row and returns this rank's rows: each rank's loss depends on each rank's rows through the batch
    statistics (SyncBatchNorm semantics, reference train_ae.py:152).  No host synchronisation: capturable."""

    @staticmethod
    def forward(ctx, t, lay):
        ctx.lay = lay
        return _gather_padded(t, lay)

    @staticmethod
    def backward(ctx, g_all):
        lay = ctx.lay
        g_all = g_all.contiguous().clone()
        run(dist.all_reduce, g_all, op=dist.ReduceOp.SUM)
        return g_all[lay.row0:lay.row0 + lay.sizes[dist.get_rank()]], None


def gather_rows(t):
    """-> (rows of all ranks (differentiable), RowLayout).  Identity without a process group."""
    lay = row_layout(t.shape[0], t.device)
    if len(lay.sizes) == 1 and not sharded():
        return t, lay
    return GatherRows.apply(t, lay), lay


def all_reduce_direct(flat, group=None):
    """Sum `flat` over the ranks with the DIRECT algorithm: reduce-scatter + all-gather as messages between every pair of
    ranks, not around a ring.  On an MI355X node every GPU has its own xGMI link to each of the other seven (7 x ~153 GB/s, no
    switch), so a rank sends its W - 1 chunks to their owners at the same time over W - 1 different links and the exchange is two
    steps of n / W elements per link -- a ring all-reduce moves 2 (W - 1) / W n elements through EVERY link in 2 (W - 1) dependent
    steps.  Step 1 is ONE all_to_all_single (chunk r of every rank to rank r: RCCL runs it as grouped sends / receives between all
    pairs), the owner adds the W chunks in rank order (the same sum on every run), step 2 one all_gather_into_tensor.  Both are plain
    stream-ordered collectives: no Work object, no host wait -- CAPTURABLE in a hipGraph like the ring all-reduce (round 4's
    batch_isend_irecv + wait() form was not).  In place; `flat`: contiguous 1-D, any length (padded to W chunks on the fly)."""
    world = dist.get_world_size(group)
    if world == 1:
        return flat
    n = flat.numel()
    chunk = (n + world - 1) // world
    if chunk * world == n:
        send = flat
    else:
        send = flat.new_zeros(chunk * world)
        send[:n].copy_(flat)
    inbox = torch.empty_like(send)
    kw = {'group': group} if group is not None else {}
    run(dist.all_to_all_single, inbox, send, **kw)                       # row r of inbox: rank r's copy of MY chunk
    mine = inbox.view(world, chunk).sum(0)                               # fixed order: rank 0's, rank 1's, ...
    if dist.get_backend(group) == 'nccl':
        run(dist.all_gather_into_tensor, send, mine, **kw)
    else:
        run(dist.all_gather, list(send.chunk(world)), mine, **kw)
    if send is not flat:
        flat.copy_(send[:n])
    return flat


def _sum_over_ranks(flat, algorithm):
    """In-place sum of a flat buffer over the default group: 'ring' = the library's all-reduce (RCCL picks its own schedule), 'direct'
    = all_reduce_direct (sized for the xGMI mesh).  Both stream-ordered and capturable."""
    if algorithm == 'direct':
        all_reduce_direct(flat)
    elif algorithm == 'ring':
        run(dist.all_reduce, flat, op=dist.ReduceOp.SUM)
    else:
        raise ValueError(f'unknown gradient exchange algorithm {algorithm!r}')


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
    _sum_over_ranks(flat, algorithm)
    if average:
        flat.div_(dist.get_world_size())
    # scatter back with ONE multi-tensor copy (a per-parameter copy_ is ~650 launches for the airplane model)
    torch._foreach_copy_(grads, [v.view_as(g) for v, g in zip(flat.split_with_sizes([g.numel() for g in grads]), grads)])
    return flat.numel()


def sync_module_state(module, src=0, verify_only=False):
    """Make every rank start from rank `src`'s model: broadcast all parameters and buffers (BatchNorm running statistics,
    num_batches_tracked, eps).  DistributedDataParallel does this at construction (reference train_ae.py:153) and the reference
    sets no seed, so without it ranks that built the model themselves -- or where only rank 0 loaded a checkpoint -- would
    average the gradients of DIFFERENT models, silently.  One flat broadcast per dtype.  verify_only: change nothing, raise if a
    rank's state differs from `src`'s (a checksum exchange).  No-op without a process group or on one rank.
    Returns the number of elements broadcast / checked."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    tensors = [p.data for p in module.parameters()] + [b for b in module.buffers()]
    on_cpu = dist.get_backend() != 'nccl'
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    total = 0
    for dtype in sorted(by_dtype, key=str):                   # the same order on every rank
        group = by_dtype[dtype]
        flat = torch.cat([t.detach().reshape(-1) for t in group])
        wire = flat.cpu() if (on_cpu and flat.is_cuda) else flat.clone()
        run(dist.broadcast, wire, src=src)
        wire = wire.to(flat.device)
        total += flat.numel()
        if verify_only:
            same = torch.equal(wire, flat) or bool(((wire == flat) | (torch.isnan(wire.float()) & torch.isnan(flat.float()))).all())
            ok = torch.tensor([1 if same else 0], dtype=torch.int64, device=wire.device if not on_cpu else 'cpu')
            run(dist.all_reduce, ok, op=dist.ReduceOp.MIN)
            if int(ok) == 0:
                raise RuntimeError(f"rank {dist.get_rank()}: the model's {dtype} state differs between the ranks "
                                   f'(parameters / buffers not synchronised: call dist.sync_module_state(model) first)')
            continue
        with torch.no_grad():
            torch._foreach_copy_(group, [v.view_as(t) for v, t in zip(wire.split_with_sizes([t.numel() for t in group]), group)])
    if not verify_only:
        for mod in module.modules():                           # every packed-weight cache is keyed on the tensors just overwritten
            if hasattr(mod, 'invalidate_packed_weights'):
                mod.invalidate_packed_weights()
    return total


_REDUCE_STREAMS = {}        # per device: the stream the overlapped gradient all-reduces are ordered on


class OverlappedGradients:
    """Data-parallel gradient exchange overlapped with the backward pass (the job DistributedDataParallel's buckets do in
    the reference, train_ae.py:153).  The decoders hold ~3/4 of the model's parameters (3.7 M of 4.8 M for the airplane
    config) and are the LAST modules of the forward pass, so their gradients are complete FIRST in the backward pass -- one
    flat tensor per decoder (the gradient of its raw parameter arena).  Their all-reduces are launched asynchronously
    right there (on a side stream: RCCL runs over xGMI) while the encoder / prior-flow backward continues on the compute
    stream; ``finish()`` reduces what is left as one more flat buffer and waits.  Capturable in a hipGraph.

        reducer = OverlappedGradients(model)          # once
        with reducer:                                  # per step; the forward pass may be inside or outside
            loss.backward()
        optimizer.step()

    Contract: a decoder's parameters receive gradient only through its raw arena (true for every path of this package), so the
    reduced flat gradient REPLACES the .grad this backward pass created.  Gradient accumulation: run the earlier micro-batches
    outside the reducer (DistributedDataParallel.no_sync semantics) and the last one inside -- decoders whose parameters already
    hold a .grad are then reduced with the remainder, as accumulated totals (tests/dist_gpu_worker.py).
    """

    def __init__(self, module, average=True, sync_state=True, algorithm=None):
        """algorithm: 'ring' (the library all-reduce) or 'direct' (all_reduce_direct: one all-to-all + one all-gather over the xGMI
        mesh) for EVERY exchange of this reducer -- the per-decoder ones launched from inside the backward pass and the remainder --
        so that an 8-GPU run can A/B the two inside the captured step; default: GWTF_GRAD_ALGORITHM or 'ring'."""
        self.module, self.average = module, average
        self.algorithm = algorithm or os.environ.get('GWTF_GRAD_ALGORITHM', 'ring')
        if self.algorithm not in ('ring', 'direct'):
            raise ValueError(f'unknown gradient exchange algorithm {self.algorithm!r}')
        self.pending = []
        self.launched = 0          # asynchronous collectives launched from inside the backward pass (tests read it)
        # every rank starts from rank 0's parameters AND buffers, as DistributedDataParallel's constructor guarantees
        # (train_ae.py:153); sync_state=False: the caller has done it (verify with sync_module_state(verify_only=True))
        self.synced = sync_module_state(module) if sync_state else 0

    def __enter__(self):
        from . import autograd
        self.pending = []
        autograd.GRAD_SINK['reducer'] = self if sharded() else None
        return self

    def __exit__(self, exc_type, exc, tb):
        from . import autograd
        autograd.GRAD_SINK['reducer'] = None
        if exc_type is None:
            self.finish()
        return False

    def on_flat_gradient(self, grad, engine):
        """Called from the backward pass with one decoder's flat gradient (the gradient of its raw arena; the parameters' .grad
        are about to be made from slices of it).  The all-reduce runs on a PRIVATE copy: the compute stream goes on reading
        `grad` (AccumulateGrad) while RCCL works, so reducing it in place would be a data race.
        Gradient accumulation: when a covered parameter already holds a .grad (earlier micro-batches, run OUTSIDE the reducer
        like DistributedDataParallel.no_sync), what must be reduced is the accumulated total, which exists only after this
        backward pass -- such a decoder is left to finish()'s flat buffer."""
        if any(t is not None and t.grad is not None for t, _op in engine._srcs):
            return
        buf = grad.contiguous().clone()
        if not buf.is_cuda:
            _sum_over_ranks(buf, self.algorithm)
            self.pending.append((buf, engine, None))
        else:
            # a stream-ordered all-reduce on a SIDE stream (not async_op=True: a Work object created inside a hipGraph capture
            # ends up in ProcessGroupNCCL's watchdog list and its captured events cannot be queried): the collective and the
            # rest of the backward pass are parallel branches -- of the stream DAG when eager, of the graph when captured
            cur = torch.cuda.current_stream(buf.device)
            side = _REDUCE_STREAMS.get(buf.device)
            if side is None:
                side = _REDUCE_STREAMS[buf.device] = torch.cuda.Stream(device=buf.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                _sum_over_ranks(buf, self.algorithm)
            buf.record_stream(side)
            self.pending.append((buf, engine, side))
        self.launched += 1

    def finish(self):
        """Wait for the asynchronous all-reduces and hand the reduced gradients to the parameters: every covered parameter's
        .grad is REBOUND to its slice of the reduced flat buffer (what DistributedDataParallel calls gradient_as_bucket_view) --
        no copy.  (Copying the slices back into the .grad tensors autograd had made was ~70 multi-tensor launches per step for
        the 5280 decoder tensors of the airplane model: 0.35 ms.)  Inside a captured hipGraph the buffers live in the graph's
        pool, so the rebinding done at capture time stays valid for every replay."""
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        covered = set()
        for buf, e, side in self.pending:
            if side is not None:
                torch.cuda.current_stream(buf.device).wait_stream(side)
            if self.average:
                buf.div_(world)
            for (t, _op), v in zip(e._srcs, buf.split_with_sizes(e._sizes)):
                if t is None or not t.requires_grad or t.grad is None:
                    continue
                covered.add(id(t))                # .grad was created by this backward pass: it IS the local gradient -> replace it
                t.grad = v.view_as(t)
        self.pending = []
        if world == 1 and not sharded():
            return
        rest = [p for p in self.module.parameters() if p.grad is not None and id(p) not in covered]
        if rest:
            flat = torch.cat([p.grad.reshape(-1) for p in rest])
            _sum_over_ranks(flat, self.algorithm)
            if self.average:
                flat.div_(world)
            for p, v in zip(rest, flat.split_with_sizes([p.grad.numel() for p in rest])):
                p.grad = v.view_as(p)
