"""K flow components as one launch + the fused mixture NLL.

The reference loops over ``self.pc_decoder[i]`` in Python (lib/networks/flow_mixture.py:163-166: K sequential
decoders on the SAME points when training; the N points split among the components by a multinomial draw when
sampling, :146-177) and then over B x K in ``FlowMixtureNLL`` (lib/networks/losses.py:109-131).  Here the K
components' packed weights are concatenated, one FiLM launch covers all K*C couplings, one stack launch covers
all components, and one reduction kernel produces the per-shape NLL.
"""
import torch

from . import _lib


class MixtureStack:
    """Batched driver for an ``nn.ModuleList`` of ``LocalCondRNVPDecoder`` with identical (n_flows, f, G)."""

    def __init__(self, decoders):
        self.decoders = list(decoders)
        self.engines = [d.engine() for d in self.decoders]
        e0 = self.engines[0]
        if any((e.C, e.f, e.G, e.pattern0) != (e0.C, e0.f, e0.G, e0.pattern0) for e in self.engines):
            raise ValueError('all mixture components must share n_flows, f_n_features and g_n_features')
        self.K, self.C, self.f, self.G = len(self.engines), e0.C, e0.f, e0.G
        self._cat_key, self._cat = None, None

    def __reduce__(self):
        """copy.deepcopy / pickle: a fresh stack over the (copied) decoders -- the caches here belong to the original's tensors."""
        return (MixtureStack, (self.decoders,))

    def packed(self):
        packs = [e.packed(False) for e in self.engines]
        key = tuple(id(pk[0]) for pk in packs)
        if key != self._cat_key:
            self._cat = (torch.cat([pk[0] for pk in packs]), torch.cat([pk[1] for pk in packs]))
            self._cat_key, self._keep = key, packs
        return self._cat

    def packed_exact(self):
        """The K components' exact-fp32 operand records, concatenated (re-run of out-of-range tiles, flows.range_rerun)."""
        self.packed()
        pxs = [e.packed_exact() for e in self.engines]
        key = tuple(id(x) for x in pxs)
        if key != getattr(self, '_catx_key', None):
            # (each engine's record carries its own work list behind it, _lib.pack_weights_exact: the K record parts, then ONE list)
            n = self.C * _lib.lib().gwtf_packed_x_coupling_floats(self.f)
            tail = torch.zeros(_lib.WORKLIST_INTS, device=pxs[0].device, dtype=torch.float32)
            self._catx, self._catx_key, self._keepx = torch.cat([x[:n] for x in pxs] + [tail]), key, pxs
        return self._catx

    def _film(self, g):
        pw, pf = self.packed()
        eps = self.engines[0].couplings[0]._eps_value
        return pw, _lib.film_forward(g, pf, self.K * self.C, self.f, eps, False), eps

    def forward_all(self, p, g, mode='inverse'):
        """Every component on every point -> (out, logdet), each (K,B,3,N).  Training / density path."""
        e0 = self.engines[0]
        e0._check(p, g)
        needs_grad = torch.is_grad_enabled() and (p.requires_grad or g.requires_grad or any(
            t.requires_grad for d in self.decoders for t in d.parameters()))
        if e0.couplings[0].training and not any(getattr(e, 'force_autograd_chain', False) for e in self.engines):
            # batch-statistic BatchNorm: all K components through every kernel of the train pipeline together
            # (csrc/gwtf_train.hip, K-batched pipeline); data-parallel runs all-reduce one packed statistic per phase
            import torch.distributed as dist
            from .autograd import train_density_forward_multi
            from .flows import _sharded
            multi = _sharded()
            with torch.set_grad_enabled(needs_grad):
                out, logdet, _, bn_batch = train_density_forward_multi(self.engines, p, g, mode, distributed=multi, want_lists=False)
            for k, e in enumerate(self.engines):
                e._update_running_stats(bn_batch[k])
                e._last_lists = None
            return out, logdet
        if e0.couplings[0].training or needs_grad:
            # eval BatchNorm with autograd (or the cross-check chain): the per-component differentiable path
            res = [e.run(p, g, mode, False) for e in self.engines]
            return torch.stack([r[0] for r in res]), torch.stack([r[1] for r in res])
        from .flows import range_rerun
        pw, film, eps = self._film(g.contiguous().float())
        px = self.packed_exact() if (range_rerun() or _lib.EXACT[0]) else None
        return _lib.stack_forward_multi(p.contiguous().float(), pw, film, self.K, self.C, self.f, e0.pattern0, eps, mode, packed_x=px)

    def forward_all_lists(self, p, g, mode='inverse', defer_running_stats=False):
        """Every component on every point, train-mode BatchNorm, WITH the reference's per-coupling lists: -> (out, logdet (K,B,3,N),
        (ps, mus, logvars) each (K, C, B, 3, N) direct-ordered; ps / logvars differentiable in every slot, a gradient through mus
        raises) -- the K list-API decoder calls of flow_mixture.py:163-166 as ONE pass of the K-batched pipeline.  None when that
        pipeline does not apply (eval-mode BatchNorm, the cross-check chain): the caller then takes the per-decoder route.
        defer_running_stats: return bn_batch (K, ...) as a fourth value instead of updating the BatchNorm buffers here."""
        e0 = self.engines[0]
        e0._check(p, g)
        if not e0.couplings[0].training or any(getattr(e, 'force_autograd_chain', False) for e in self.engines):
            return None
        from .autograd import train_density_forward_multi
        from .flows import _sharded
        needs_grad = torch.is_grad_enabled() and (p.requires_grad or g.requires_grad or any(
            t.requires_grad for d in self.decoders for t in d.parameters()))
        with torch.set_grad_enabled(needs_grad):
            out, logdet, lists, bn_batch = train_density_forward_multi(self.engines, p, g, mode, distributed=_sharded())
        if defer_running_stats:
            # the sibling round of decoders._SiblingGroup: every decoder's buffers are updated when its own call arrives
            return out, logdet, lists, bn_batch
        for k, e in enumerate(self.engines):
            e._update_running_stats(bn_batch[k])
            e._last_lists = None
        return out, logdet, lists

    def forward_partition(self, p, g, counts, mode='direct'):
        """Sampling path: the N points are laid out component by component, ``counts[k]`` points for component k
        (sum == N); each point goes through ONE component.  -> (out, logdet), each (B,3,N)."""
        e0 = self.engines[0]
        if e0.couplings[0].training:
            raise NotImplementedError('the sampling partition is an evaluation path (reference flow_mixture.py:146); call .eval()')
        e0._check(p, g)
        if len(counts) != self.K or sum(int(c) for c in counts) != p.shape[2]:
            raise ValueError('counts must have K entries summing to N')
        segs, off = [], 0
        for cnt in counts:
            segs.append((off, off + int(cnt)))
            off += int(cnt)
        pw, film, eps = self._film(g.contiguous().float())
        # no re-run launch here (base samples of a few units; one shape per call is latency-bound) unless the exact body is forced
        px = self.packed_exact() if _lib.EXACT[0] else None
        return _lib.stack_forward_multi(p.contiguous().float(), pw, film, self.K, self.C, self.f, e0.pattern0, eps, mode,
                                        segments=segs, shared_points=False, packed_x=px)


class _MixtureNLLFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, logdet, mu0, lv0, logits):
        z, logdet, mu0, lv0, logits = (t.contiguous().float() for t in (z, logdet, mu0, lv0, logits))
        nll, plse = _lib.mixture_nll(z, logdet, mu0, lv0, logits, True)
        ctx.save_for_backward(z, logdet, mu0, lv0, logits, plse)
        ctx.mark_non_differentiable(plse)
        return nll, plse

    @staticmethod
    def backward(ctx, g_nll, _g_plse):
        z, logdet, mu0, lv0, logits, plse = ctx.saved_tensors
        L = _lib.lib()
        K, B, _, N = z.shape
        g_z, g_ld = torch.empty_like(z), torch.empty_like(logdet)
        g_mu0, g_lv0, g_logits = torch.empty_like(mu0), torch.empty_like(lv0), torch.empty_like(logits)
        g_nll = g_nll.contiguous().float()
        with torch.cuda.device(z.device):
            _lib.check(L.gwtf_mixture_nll_backward(z.data_ptr(), logdet.data_ptr(), mu0.data_ptr(), lv0.data_ptr(),
                                                   logits.data_ptr(), plse.data_ptr(), g_nll.data_ptr(), g_z.data_ptr(),
                                                   g_ld.data_ptr(), g_mu0.data_ptr(), g_lv0.data_ptr(), g_logits.data_ptr(),
                                                   K, B, N, _lib._stream(z)))
        return g_z, g_ld, g_mu0, g_lv0, g_logits


def flow_mixture_nll(z, logdet, mu0, lv0, logits, want_point_lse=False):
    """FlowMixtureNLL on fused decoder outputs (reference losses.py:88-137).

    z, logdet (K,B,3,N): final inverse coordinates / sum of coupling logvars per component;
    mu0, lv0 (K,B,3): base Gaussians; logits (B,K).  -> (pnll scalar = batch mean, per-shape (B,) [, per-point lse])."""
    nll, plse = _MixtureNLLFn.apply(z, logdet, mu0, lv0, logits)     # differentiable w.r.t. all five inputs
    return (nll.mean(), nll, plse) if want_point_lse else (nll.mean(), nll)
