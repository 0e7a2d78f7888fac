"""Conditional affine couplings of the discrete point flow, MI355X-native.

Same constructor signatures, attribute names, ``state_dict`` keys and ``forward(p, g, mode)``
contract as the reference's ``lib/networks/flows.py`` (``CondRealNVPFlow3D`` :10-117,
``CondRealNVPFlow3DTriple`` :120-160).  The sub-modules below only HOLD parameters and
BatchNorm buffers under the reference's names; they are never called.  All arithmetic runs in
hand-written HIP kernels (``csrc/``) through the C ABI in ``include/gwtf.h``: one weight-pack
launch (cached while parameters are unchanged), one FiLM launch, one fused stack launch.
There is no torch/CPU fallback: tensors must live on a HIP device.
"""
import operator
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib
from .layers import SharedDot, Swish

_VERSION = operator.attrgetter('_version')

# Points whose coordinates leave the split-f16 contraction's operand range (|x| > 3e4, csrc/gwtf_layout.h GWTF_X_LIMIT) are flagged
# NaN by the stack kernel; the eval-mode density / list paths follow every stack launch with the exact-fp32 re-run launch
# (csrc/gwtf_stack_exact.hip: tiles without a flagged point leave at once, ~2 us), so such points come back with the
# reference's finite values (reference flows.py:113-115 has no range limit).  GWTF_NO_RANGE_RERUN=1 switches it off (NaN as in
# rounds 1-4).  The sampling partition (MixtureStack.forward_partition: base samples, |z| of a few units, latency-bound at one
# shape per call) and the train pipeline do not re-run.
import os as _os


def range_rerun():
    return _os.environ.get('GWTF_NO_RANGE_RERUN') != '1'

# index -> warped coordinates; couplings of a decoder cycle through these (reference flows.py:129-148)
WARP_PATTERNS = ((0,), (1,), (2,), (0, 1), (0, 2), (1, 2))


def _sharded():
    """dist.sharded(): BatchNorm statistics are summed over the ranks of torch.distributed's default group."""
    from .dist import sharded
    return sharded()


def _graph_capture(graph):
    from .dist import graph_capture
    return graph_capture(graph)


def _film_head(X, which, f, G):
    n = f'{X}_sd1_film_{which}'
    return nn.Sequential(OrderedDict([
        (n + '0', nn.Linear(G, f, bias=False)),
        (n + '0_bn', nn.BatchNorm1d(f)),
        (n + '0_swish', Swish()),
        (n + '1', nn.Linear(f, f, bias=True)),
    ]))


class CondRealNVPFlow3D(nn.Module):
    """One elementary coupling: kept coordinates -> (mu, logvar) of the warped ones, conditioned on g."""

    def __init__(self, f_n_features, g_n_features, weight_std=0.01, warp_inds=[0],
                 centered_translation=False, eps=1e-6):
        super().__init__()
        self.f_n_features, self.g_n_features = f_n_features, g_n_features
        self.weight_std = weight_std
        self.warp_inds = list(warp_inds)
        if tuple(self.warp_inds) not in WARP_PATTERNS:
            raise ValueError(f'warp_inds {warp_inds} is not one of {WARP_PATTERNS}')
        self.keep_inds = [d for d in (0, 1, 2) if d not in self.warp_inds]
        self.centered_translation = centered_translation  # stored, never used (as in the reference)
        self.register_buffer('eps', torch.tensor([eps], dtype=torch.float32))
        self._eps_value = float(eps)
        f, G, k, w = f_n_features, g_n_features, len(self.keep_inds), len(self.warp_inds)
        for X in ('mu', 'logvar'):
            setattr(self, f'T_{X}_0', nn.Sequential(OrderedDict([
                (f'{X}_sd0', SharedDot(k, f, 1)),
                (f'{X}_sd0_bn', nn.BatchNorm1d(f)),
                (f'{X}_sd0_relu', nn.ReLU(inplace=True)),
                (f'{X}_sd1', SharedDot(f, f, 1)),
                (f'{X}_sd1_bn', nn.BatchNorm1d(f, affine=False)),
            ])))
            setattr(self, f'T_{X}_0_cond_w', _film_head(X, 'w', f, G))
            setattr(self, f'T_{X}_0_cond_b', _film_head(X, 'b', f, G))
            setattr(self, f'T_{X}_1', nn.Sequential(OrderedDict([
                (f'{X}_sd1_relu', nn.ReLU(inplace=True)),
                (f'{X}_sd2', SharedDot(f, w, 1, bias=True)),
            ])))
            # near-identity start: last layers ~ N(0, weight_std), zero bias (reference flows.py:52-58,87-93)
            with torch.no_grad():
                for head in (getattr(self, f'T_{X}_0_cond_w')[-1], getattr(self, f'T_{X}_0_cond_b')[-1],
                             getattr(self, f'T_{X}_1')[-1]):
                    head.weight.normal_(std=weight_std)
                    head.bias.zero_()
        self._engine = None
        self._stamp = 0

    # -- parameter gathering -------------------------------------------------------------------
    def raw_sources(self):
        """This coupling's record of the raw arena as (tensor, op) pairs in the order fixed by csrc/gwtf_layout.h;
        op: None = flatten as is, int n = n zeros (padding of the absent kept / warped slot)."""
        out = []
        k, w = len(self.keep_inds), len(self.warp_inds)
        f = self.f_n_features
        for X in ('logvar', 'mu'):
            t0 = getattr(self, f'T_{X}_0')
            sd0, bn0, sd1, bn1 = t0[0], t0[1], t0[3], t0[4]
            out += [(sd0.weight, None)] + ([(None, f)] if k < 2 else [])
            out += [(bn0.weight, None), (bn0.bias, None), (bn0.running_mean, None), (bn0.running_var, None),
                    (sd1.weight, None), (bn1.running_mean, None), (bn1.running_var, None)]
            for which in ('w', 'b'):
                head = getattr(self, f'T_{X}_0_cond_{which}')
                out += [(head[0].weight, None), (head[1].weight, None), (head[1].bias, None),
                        (head[1].running_mean, None), (head[1].running_var, None), (head[3].weight, None),
                        (head[3].bias, None)]
            sd2 = getattr(self, f'T_{X}_1')[1]
            out += [(sd2.weight, None)] + ([(None, f)] if w < 2 else [])
            out += [(sd2.bias, None)] + ([(None, 1)] if w < 2 else [])
        return out

    def raw_tensors(self):
        """Flattened tensors of this coupling's raw-arena record (see raw_sources)."""
        dev = self.eps.device
        return [torch.zeros(op, device=dev) if t is None else t.reshape(-1)
                for t, op in self.raw_sources()]

    def tracked_tensors(self):
        """Every tensor whose in-place modification must invalidate the packed-weight cache."""
        return [t for t in list(self.parameters()) + list(self.buffers()) if t.dtype == torch.float32]

    # -- packed-weight cache invalidation ---------------------------------------------------------
    # The reference optimiser writes through ``p.data`` (lib/networks/optimizers.py:69-72), which does
    # not bump tensor version counters, so besides versions the cache is keyed on a stamp bumped by
    # everything a training script does between two eval forwards: train()/eval(), .to()/.cuda(),
    # load_state_dict().  Call ``invalidate_packed_weights()`` after any other out-of-band edit.
    def invalidate_packed_weights(self):
        self._stamp = getattr(self, '_stamp', 0) + 1

    def train(self, mode=True):
        self.invalidate_packed_weights()
        return super().train(mode)

    def _apply(self, fn, *args, **kwargs):
        self.invalidate_packed_weights()
        return super()._apply(fn, *args, **kwargs)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        self.invalidate_packed_weights()
        if prefix + 'eps' in state_dict:
            self._eps_value = float(state_dict[prefix + 'eps'].reshape(-1)[0])

    def forward(self, p, g, mode='direct'):
        """-> (p_out, mu, logvar), each (B,3,N); reference flows.py:95-117."""
        if self._engine is None:
            self._engine = StackEngine([self])
        ps, mus, lvs = self._engine.run_lists(p, g, mode)
        return ps[0], mus[0], lvs[0]


class CondRealNVPFlow3DTriple(nn.Module):
    """Three couplings warping [0],[1],[2] (pattern 0) or [0,1],[0,2],[1,2] (pattern 1)."""

    def __init__(self, f_n_features, g_n_features, weight_std=0.02, pattern=0, centered_translation=False):
        super().__init__()
        if pattern not in (0, 1):
            raise ValueError(f'pattern must be 0 or 1, got {pattern}')
        self.f_n_features, self.g_n_features = f_n_features, g_n_features
        self.weight_std, self.pattern, self.centered_translation = weight_std, pattern, centered_translation
        for j in range(3):
            setattr(self, f'nvp{j + 1}', CondRealNVPFlow3D(
                f_n_features, g_n_features, weight_std=weight_std, warp_inds=list(WARP_PATTERNS[3 * pattern + j]),
                centered_translation=centered_translation))
        self._engine = None

    def couplings(self):
        return [self.nvp1, self.nvp2, self.nvp3]

    def forward(self, p, g, mode='direct'):
        """-> ([p1,p2,p3], [mu1..3], [logvar1..3]) in nvp1..3 order for both modes; reference flows.py:150-160."""
        if self._engine is None:
            self._engine = StackEngine(self.couplings())
        return self._engine.run_lists(p, g, mode)


class _ArenaCat(torch.autograd.Function):
    """raw arena = cat(flattened parameters / buffers) as ONE autograd node: forward concatenates cached detached views,
    backward hands every parameter its slice of the flat gradient (views, no kernels)."""

    @staticmethod
    def forward(ctx, engine, *tensors):
        ctx.engine = engine
        return engine._cat_detached()

    @staticmethod
    def backward(ctx, g_raw):
        eng = ctx.engine
        pieces = g_raw.split_with_sizes(eng._sizes)
        grads, k = [], 0
        for (t, op), piece in zip(eng._srcs, pieces):
            if t is None:
                continue
            k += 1
            if not ctx.needs_input_grad[k]:
                grads.append(None)
            else:
                grads.append(piece.view(t.shape))
        return (None, *grads)


class _ArenaCatMulti(torch.autograd.Function):
    """The raw arenas of K engines of one shape as ONE (K, R) tensor built by ONE pointer-table launch (a mixture's K decoders:
    instead of K gathers and a torch.stack); backward hands every parameter its slice, like _ArenaCat."""

    @staticmethod
    def forward(ctx, engines, *tensors):
        ctx.engines = engines
        return _gather_stacked(engines)

    @staticmethod
    def backward(ctx, g_raw):
        grads = []
        for k, eng in enumerate(ctx.engines):
            pieces = g_raw[k].split_with_sizes(eng._sizes)
            for (t, op), piece in zip(eng._srcs, pieces):
                if t is not None:
                    grads.append(piece.view(t.shape) if t.requires_grad else None)
        return (None, *grads)


def _gather_stacked(engines):
    """(K, R) stacked raw arena from the engines' cached detached views; the combined device table is cached per engine tuple and
    rebuilt when a source pointer changed (not inside a hipGraph capture: per-engine gathers + torch.stack stand in there)."""
    for e in engines:
        e._refresh_flat()
    flat0 = engines[0]._flat
    R = sum(engines[0]._sizes)
    if not flat0[0].is_cuda:
        return torch.stack([torch.cat(e._flat) for e in engines])
    # the combined table lives on the first engine (it dies with the model: a module-level cache keyed on id() would keep the
    # detached parameter views and the device table of every model ever built alive)
    key = tuple(id(e) for e in engines)
    tables = engines[0].__dict__.setdefault('_stack_tables', {})
    tab = tables.get(key)
    flats = [e._flat for e in engines]
    if tab is None or len(tab[0]) != len(flats) or any(x is not y for x, y in zip(tab[0], flats)):   # a flat list is rebuilt with its tensors
        rows = []
        for k, e in enumerate(engines):
            off = k * R
            for t, n in zip(e._flat, e._sizes):
                rows.append((t.data_ptr(), off, n))
                off += n
        if tab is not None and tab[1] == rows:
            tab = tables[key] = (flats, rows, tab[2])
        elif torch.cuda.is_current_stream_capturing():
            return torch.stack([e._cat_detached() for e in engines])
        else:
            tab = tables[key] = (flats, rows, torch.tensor(rows, dtype=torch.int64).to(flat0[0].device))
    out = torch.empty(len(engines), R, device=flat0[0].device, dtype=torch.float32)
    with torch.cuda.device(out.device):
        _lib.check(_lib.lib().gwtf_gather_table(tab[2].data_ptr(), out.data_ptr(), len(tab[1]), _lib._stream(out)))
    return out


def stacked_raw_arena(engines):
    """(K, R): the raw arenas of K same-shaped engines, differentiable w.r.t. every parameter (one autograd node)."""
    engines = list(engines)
    for e in engines:
        e._refresh_flat()
    srcs = [t for e in engines for t, _ in e._srcs if t is not None]
    if torch.is_grad_enabled() and any(t.requires_grad for t in srcs):
        return _ArenaCatMulti.apply(engines, *srcs)
    return _gather_stacked(engines)


class StackEngine:
    """Host-side driver of the HIP path for a run of consecutive couplings (direct order).

    Gathers the couplings' parameters into the raw arena, packs them (cached until a parameter or
    buffer changes), runs the FiLM kernel and the fused stack kernel."""

    def __init__(self, couplings):
        self.couplings = list(couplings)
        c0 = self.couplings[0]
        self.C = len(self.couplings)
        self.f, self.G = c0.f_n_features, c0.g_n_features
        pats = [WARP_PATTERNS.index(tuple(c.warp_inds)) for c in self.couplings]
        self.pattern0 = pats[0]
        if any(pt != (self.pattern0 + i) % 6 for i, pt in enumerate(pats)):
            raise ValueError(f'couplings do not follow the cyclic warp pattern: {pats}')
        if self.f > 128:
            raise NotImplementedError(f'f_n_features={self.f} > 128 is not supported by the gfx950 kernels '
                                      '(reference configs use <= 64; the forward kernels reach 128, train / backward 96)')
        self._tracked, self._tracked_stamp = [], None
        self._srcs, self._src_stamp, self._zeros = None, None, None
        self._cache_key = None
        self._packed = None
        self._flat_key, self._flat = None, None
        self._bn_cache = None

    def __reduce__(self):
        """copy.deepcopy / pickle of a module that owns an engine: a FRESH engine over the (copied) couplings.  Every cache in
        here -- detached views of the parameters, device pointer tables, packed weights -- describes the original's tensors; a
        field-by-field copy would keep reading them (a deep copy of a model taken after it ran would ignore its own weights)."""
        return (StackEngine, (self.couplings,))

    def raw_arena(self):
        """Parameters + BatchNorm buffers of all couplings as one flat tensor (autograd-aware torch.cat).  The list
        of sources is cached (module traversal costs more than the copy); it is rebuilt when .to()/.cuda()/
        load_state_dict() re-create tensors (stamp change)."""
        self._refresh_flat()
        if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t, _ in self._srcs):
            return _ArenaCat.apply(self, *[t for t, _ in self._srcs if t is not None])
        return self._cat_detached()

    def _refresh_flat(self):
        """(Re)build the cached source list and its detached flat views."""
        stamp = sum(c._stamp for c in self.couplings)
        if stamp != self._src_stamp:
            self._srcs = [so for c in self.couplings for so in c.raw_sources()]
            dev = self.couplings[0].eps.device
            self._zeros = {n: torch.zeros(n, device=dev) for n in {op for t, op in self._srcs if t is None}}
            self._src_stamp = stamp
            self._flat_key = None
        # The 1320 flattened views (11-Triple decoder) are cached as DETACHED views: building them costs more host time
        # than the copy.  They must not carry autograd history -- a cached differentiable view of a parameter that the
        # optimiser then updates in place gets an AsStridedBackward grad_fn (new_zeros + copy per parameter, measured:
        # +2000 kernels per step).  The autograd link is one custom node instead (_ArenaCat below).
        if self._flat_key is None:
            z = self._zeros
            self._flat = [z[op] if t is None else t.detach().view(-1) for t, op in self._srcs]
            self._sizes = [op if t is None else t.numel() for t, op in self._srcs]
            self._flat_key = True

    def _cat_detached(self):
        """The flat arena from the cached detached views: one pointer-table kernel (csrc/gwtf_train.hip) on a HIP device --
        torch.cat would be a launch per 128 inputs, eleven for an 11-Triple decoder.  The device table is rebuilt only when a
        source pointer changed (it is a host-to-device copy: not allowed inside a hipGraph capture, where torch.cat stands in)."""
        flat = self._flat
        if not flat[0].is_cuda:
            return torch.cat(flat)
        rows, off = [], 0
        tab = getattr(self, '_gather_tab', None)
        if tab is None or tab[0] is not flat:
            for t, n in zip(flat, self._sizes):
                rows.append((t.data_ptr(), off, n))
                off += n
            if tab is not None and tab[1] == rows:
                tab = self._gather_tab = (flat, rows, tab[2], off)
            elif torch.cuda.is_current_stream_capturing():
                return torch.cat(flat)
            else:
                tab = self._gather_tab = (flat, rows, torch.tensor(rows, dtype=torch.int64).to(flat[0].device), off)
        _, rows, table, total = tab
        out = torch.empty(total, device=flat[0].device, dtype=torch.float32)
        with torch.cuda.device(out.device):
            _lib.check(_lib.lib().gwtf_gather_table(table.data_ptr(), out.data_ptr(), len(rows), _lib._stream(out)))
        return out

    def _collect(self):
        self._key(False)
        return self._tracked

    def _key(self, training):
        stamp = sum(c._stamp for c in self.couplings)
        if stamp != self._tracked_stamp:  # buffers are re-created by .to()/.cuda(): re-collect
            self._tracked = [t for c in self.couplings for t in c.tracked_tensors()]
            self._tracked_stamp = stamp
        return (training, stamp, sum(map(_VERSION, self._tracked)))

    def packed(self, training):
        key = self._key(training)
        if key != self._cache_key:
            with torch.no_grad():
                raw = self.raw_arena()
                self._packed = _lib.pack_weights(raw, self.C, self.f, self.G, training, self.pattern0)
            self._cache_key = key
            self._packed_x = None
        return self._packed

    def packed_exact(self):
        """The exact-fp32 operand record of the CURRENT eval packing (cached with it)."""
        pw, pf = self.packed(False)
        if getattr(self, '_packed_x', None) is None:
            with torch.no_grad():
                self._packed_x = _lib.pack_weights_exact(self.raw_arena(), pf, self.C, self.f, self.G, self.pattern0)
        return self._packed_x

    def _check(self, p, g):
        if p.dim() != 3 or p.shape[1] != 3:
            raise ValueError(f'p must be (B,3,N), got {tuple(p.shape)}')
        if g.dim() != 2 or g.shape[0] != p.shape[0] or g.shape[1] != self.G:
            raise ValueError(f'g must be ({p.shape[0]},{self.G}), got {tuple(g.shape)}')
        dev = self.couplings[0].eps.device
        if dev.type != 'cuda':
            raise _lib.GwtfError('module parameters are on the CPU: move the module to a HIP device (.cuda()); '
                                 'go_with_the_flows_amd has no CPU path')
        if p.device != dev or g.device != dev:
            raise _lib.GwtfError(f'p ({p.device}) / g ({g.device}) must be on the module device {dev}')

    def run(self, p, g, mode, want_lists):
        if mode not in ('direct', 'inverse'):
            raise ValueError(f"mode must be 'direct' or 'inverse', got {mode!r}")
        self._check(p, g)
        c0 = self.couplings[0]
        if p.shape[2] == 0:
            # an empty cloud (the sampling draw of flow_mixture.py:153 gave this component no point): the reference's torch ops
            # pass empty tensors through; no launch here
            B = p.shape[0]
            lists = p.new_zeros(3, self.C, B, 3, 0, dtype=torch.float32) if want_lists else None
            return p.new_zeros(B, 3, 0, dtype=torch.float32), p.new_zeros(B, 3, 0, dtype=torch.float32), lists
        needs_grad = torch.is_grad_enabled() and (p.requires_grad or g.requires_grad or
                                                  any(t.requires_grad for t in self._tracked or self._collect()))
        if (needs_grad or c0.training) and self.f > 96:
            raise NotImplementedError(f'f_n_features={self.f}: train-mode BatchNorm and the backward pass are built for widths up to '
                                      '96 (their LDS working set exceeds 160 KiB beyond); eval-mode forward works up to 128')
        if needs_grad and c0.training:
            import torch.distributed as dist
            multi = _sharded()
            if not getattr(self, 'force_autograd_chain', False):
                # the fused pipeline: single rank = two C calls; several ranks = one packed statistic all-reduce per phase
                from .autograd import train_density_forward_fast
                out, logdet, lists, bn_batch = train_density_forward_fast(self, p, g, mode, distributed=multi)
                self._update_running_stats(bn_batch)
                self._last_lists = None
                return out, logdet, lists
            from .autograd import train_density_forward
            out, logdet, (ps, mus, lvs), bn_batch = train_density_forward(self, p, g, distributed=multi, mode=mode)
            self._update_running_stats(bn_batch)
            self._last_lists = (ps, mus, lvs)
            lists = (torch.stack([t.detach() for t in ps]), torch.stack(mus), torch.stack([t.detach() for t in lvs])) \
                if want_lists else None
            return out, logdet, lists
        if needs_grad and not c0.training:
            # differentiable density pass: HIP forward + HIP backward (autograd.py); every ps[j] / logvars[j] list entry is
            # differentiable as in the reference (decoders.py:61-79); a gradient through a mus[j] entry raises
            from .autograd import density_forward
            out, logdet, lists = density_forward(self, p, g, mode)
            self._last_lists = None
            return out, logdet, lists
        pc, gc = p.contiguous().float(), g.contiguous().float()
        eps = c0._eps_value
        if c0.training and self.f > 64:
            # the per-coupling train kernels of this path keep a feature per lane (f <= 64): wider stacks take the fused pipeline
            from .autograd import train_density_forward_fast
            with torch.no_grad():
                out, logdet, lists, bn_batch = train_density_forward_fast(self, pc, gc, mode, distributed=_sharded())
                self._update_running_stats(bn_batch)
            return out, logdet, torch.stack(lists) if want_lists else None
        if c0.training:
            out, logdet, lists = self._run_train(pc, gc, mode, want_lists)
        else:
            pw, pf = self.packed(False)
            px = self.packed_exact() if (range_rerun() or _lib.EXACT[0]) else None
            film = _lib.film_forward(gc, pf, self.C, self.f, eps, False)
            out, logdet, lists = _lib.stack_forward(pc, pw, film, self.C, self.f, self.pattern0, eps, mode, want_lists, packed_x=px)
        return out, logdet, lists                 # no-grad paths only: both differentiable cases returned above

    # -- train mode: batch-statistic BatchNorm ----------------------------------------------------------
    def _bn_modules(self):
        """The 8 BatchNorm modules of every coupling in bn_batch order: [coupling][branch lv,mu][kind 0..3]."""
        mods = []
        for c in self.couplings:
            for X in ('logvar', 'mu'):
                t0 = getattr(c, f'T_{X}_0')
                mods += [t0[1], t0[4], getattr(c, f'T_{X}_0_cond_w')[1], getattr(c, f'T_{X}_0_cond_b')[1]]
        return mods

    def _update_running_stats(self, bn_batch):
        """running = (1-m)*running + m*batch with the unbiased batch variance (torch.nn.BatchNorm1d semantics)."""
        stamp = sum(c._stamp for c in self.couplings)
        capturing = bn_batch.is_cuda and torch.cuda.is_current_stream_capturing()
        if self._bn_cache is None or self._bn_cache[0] != stamp or (self._bn_cache[2] is None and not capturing):
            # (re)derive the pointer rows; the device table is rebuilt only when a pointer really changed (load_state_dict copies
            # in place and keeps them), because building it is a host-to-device copy -- not allowed while a hipGraph is captured
            mods = self._bn_modules()
            rows, moms, touched = [], [], []
            for i, m in enumerate(mods):
                if m.track_running_stats and m.running_mean is not None:
                    if m.momentum is None:
                        raise NotImplementedError('BatchNorm momentum=None (cumulative average) is not supported by the fused '
                                                  'running-statistic update; the reference never sets it (flows.py:27-42)')
                    nbt = m.num_batches_tracked
                    rows.append((i, m.running_mean.data_ptr(), m.running_var.data_ptr(), nbt.data_ptr() if nbt is not None else 0,
                                 float(m.momentum)))
                    touched += [m.running_mean, m.running_var] + ([nbt] if nbt is not None else [])
            old = self._bn_cache
            if old is not None and old[3] == rows and old[2] is not None and old[2][0] is not None and old[2][0].device == bn_batch.device:
                self._bn_cache = (stamp, len(mods), old[2], rows, touched)
            elif capturing:
                # no device table can be built inside a capture (host-to-device copy): this capture takes the per-module route
                # below; the next call outside a capture builds the table (the condition above)
                self._bn_cache = (stamp, len(mods), None, rows, touched)
            else:
                dev = bn_batch.device
                table = torch.tensor([r[1:4] for r in rows], dtype=torch.int64).to(dev) if rows else None
                momentum = torch.tensor([r[4] for r in rows], dtype=torch.float32).to(dev) if rows else None
                index = None if len(rows) == len(mods) else torch.tensor([r[0] for r in rows], dtype=torch.int64, device=dev)
                self._bn_cache = (stamp, len(mods), (table, momentum, index), rows, touched)
        _, n_mods, dev_tabs, rows, touched = self._bn_cache
        if not rows:
            return
        with torch.no_grad():
            flat = bn_batch.reshape(n_mods, 2, self.f)
            if dev_tabs is None:      # no device table yet (first call fell inside a graph capture): plain _foreach_ updates
                mods = self._bn_modules()
                for i, _rm, _rv, _nbt, mom in rows:
                    mods[i].running_mean.mul_(1.0 - mom).add_(flat[i, 0], alpha=mom)
                    mods[i].running_var.mul_(1.0 - mom).add_(flat[i, 1], alpha=mom)
                    if mods[i].num_batches_tracked is not None:
                        mods[i].num_batches_tracked.add_(1)
                return
            table, momentum, index = dev_tabs
            if index is not None:
                flat = flat.index_select(0, index)
            flat = flat.contiguous()
            with torch.cuda.device(flat.device):
                _lib.check(_lib.lib().gwtf_bn_running_update(table.data_ptr(), _lib._ptr(flat, 'bn_batch'), _lib._ptr(momentum, 'momentum'),
                                                            table.shape[0], self.f, _lib._stream(flat)))
            torch._C._increment_version(touched)      # written through raw pointers: the packed-weight caches key on versions

    def _run_train(self, p, g, mode, want_lists):
        """model.train() forward without autograd: statistics over all B*N points, running statistics updated with the
        modules' momentum, unbiased variance (torch semantics).  One rank: everything in HIP from one C call (the FiLM heads'
        BatchNorm over the B latent rows included).  Several ranks (the reference wraps the model in SyncBatchNorm,
        train_ae.py:152): the phase-split pipeline with its packed statistic all-reduces, as the differentiable path."""
        if _sharded():
            from .autograd import train_density_forward_fast
            with torch.no_grad():
                out, logdet, lists, bn_batch = train_density_forward_fast(self, p, g, mode, distributed=True)
                self._update_running_stats(bn_batch)
            return out, logdet, (torch.stack(lists) if want_lists else None)
        with torch.no_grad():
            raw = self.raw_arena()
            out, logdet, lists, bn_batch = _lib.train_forward(p, g, raw, self.C, self.f, self.G, self.pattern0,
                                                              self.couplings[0]._eps_value, mode, want_lists)
            self._update_running_stats(bn_batch)
        return out, logdet, lists

    def capture(self, p, g, mode, want_lists=False):
        """hipGraph capture of (FiLM + fused stack) on the CURRENT packed weights and on the storage of ``p``/``g``.

        Returns a ``GraphedStack``: ``replay()`` re-runs both kernels with one graph launch and returns the SAME
        output tensors every time (static buffers: consume them before the next replay; refill ``p``/``g`` in
        place to change inputs).  Eval mode only; re-capture after any parameter change."""
        if mode not in ('direct', 'inverse'):
            raise ValueError(f"mode must be 'direct' or 'inverse', got {mode!r}")
        self._check(p, g)
        if self.couplings[0].training:
            raise NotImplementedError('graph capture is an inference feature: call .eval() first')
        if not (p.is_contiguous() and g.is_contiguous() and p.dtype == torch.float32 and g.dtype == torch.float32):
            raise ValueError('capture needs contiguous float32 p and g (their storage is baked into the graph)')
        return GraphedStack([self], p, g, mode, want_lists)

    def run_lists(self, p, g, mode):
        out, logdet, lists = self.run(p, g, mode, True)
        ps, mus, lvs = list(lists[0].unbind(0)), list(lists[1].unbind(0)), list(lists[2].unbind(0))
        if out.requires_grad and self.couplings[0].training and getattr(self, '_last_lists', None) is not None:
            dps, dmus, dlvs = self._last_lists          # train mode: every ps[j] / logvars[j] is differentiable
            self._last_lists = None
            return list(dps), list(dmus), list(dlvs)
        return ps, mus, lvs


class GraphedStack:
    """One hipGraph holding the FiLM + stack launches of one or more engines (e.g. the K mixture components)
    that read the same ``p``/``g`` storage.  Removes the per-launch host work (Python, ctypes, allocator) from
    the step: two kernels of 10-200 us each are otherwise host-bound on small batches."""

    def __init__(self, engines, p, g, mode, want_lists=False, per_engine_p=None):
        self.engines = list(engines)
        self.p, self.g, self.mode = p, g, mode
        packs = [e.packed(False) for e in self.engines]
        pxs = [e.packed_exact() if (range_rerun() or _lib.EXACT[0]) else None for e in self.engines]
        eps = self.engines[0].couplings[0]._eps_value
        ps = per_engine_p if per_engine_p is not None else [p] * len(self.engines)

        def body():
            res = []
            for e, (pw, pf), px, pk in zip(self.engines, packs, pxs, ps):
                film = _lib.film_forward(g, pf, e.C, e.f, eps, False)
                res.append(_lib.stack_forward(pk, pw, film, e.C, e.f, e.pattern0, eps, mode, want_lists, packed_x=px))
            return res

        side = torch.cuda.Stream(device=p.device)
        side.wait_stream(torch.cuda.current_stream(p.device))
        with torch.cuda.stream(side), torch.no_grad():
            body()                                   # warm-up outside capture (lazy module loading etc.)
        torch.cuda.current_stream(p.device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with _graph_capture(self.graph), torch.no_grad():
            self.results = body()
        self._keepalive = (packs, pxs)

    def replay(self):
        self.graph.replay()
        return self.results
