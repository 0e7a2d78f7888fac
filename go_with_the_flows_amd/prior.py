"""Global prior flow on the shape latent and the Gaussian losses that consume it.

Mirror of lib/networks/flows.py:163-243 (RealNVPFlow, RealNVPFlowCouple), lib/networks/decoders.py:7-38
(GlobalRNVPDecoder) and lib/networks/losses.py:24-41 (GaussianFlowNLL, GaussianEntropy): same constructors, attribute
names, ``state_dict`` keys and list-returning forward, so the reference's checkpoints and callers (models.py:137-151)
work unchanged.  The work is per SHAPE -- B rows of G latents, 14 elementary flows of two (B x G/2)(G/2 x F) GEMM pairs.
``GlobalRNVPDecoder.forward`` on a HIP device runs the WHOLE stack as one launch (csrc/gwtf_prior.hip: forward, and one
more launch for the backward; eval- and train-mode BatchNorm, every list slot differentiable) instead of the ~100 + ~200
library launches of the module-by-module evaluation; the per-module ``forward`` of ``RealNVPFlow`` / ``RealNVPFlowCouple``
(used on their own and on the CPU by the host-logic tests) stays a chain of torch ops with
the two shipped warp patterns applied as strided slices instead of the reference's index gathers.
"""
import os
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from .layers import Swish


def _as_slice(inds, n):
    """A python slice selecting exactly ``inds`` out of range(n), or None."""
    inds = list(inds)
    if not inds:
        return None
    step = inds[1] - inds[0] if len(inds) > 1 else 1
    if step <= 0:
        return None
    sl = slice(inds[0], inds[-1] + 1, step)
    return sl if list(range(n))[sl] == inds else None


class RealNVPFlow(nn.Module):
    """One affine coupling on the latent (reference flows.py:163-216)."""

    def __init__(self, n_features, g_n_features, weight_std=0.01, warp_inds=[0], eps=1e-6):
        super().__init__()
        self.n_features, self.g_n_features, self.weight_std = n_features, g_n_features, weight_std
        self.warp_inds = [int(i) for i in warp_inds]
        self.keep_inds = [i for i in range(g_n_features) if i not in set(self.warp_inds)]
        self.register_buffer('eps', torch.from_numpy(np.array([eps], dtype=np.float32)))
        self._eps_value = float(eps)          # host copy: reading the buffer would synchronise (and break hipGraph capture)
        for X in ('mu', 'logvar'):
            branch = nn.Sequential(OrderedDict([
                (f'{X}_mlp0', nn.Linear(len(self.keep_inds), n_features, bias=False)),
                (f'{X}_mlp0_bn', nn.BatchNorm1d(n_features)),
                (f'{X}_mlp0_swish', Swish()),
                (f'{X}_mlp1', nn.Linear(n_features, len(self.warp_inds), bias=True))]))
            with torch.no_grad():
                branch[-1].weight.normal_(std=weight_std)
                branch[-1].bias.zero_()
            setattr(self, f'T_{X}_0', branch)
        self._warp_sl = _as_slice(self.warp_inds, g_n_features)
        self._keep_sl = _as_slice(self.keep_inds, g_n_features)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        if prefix + 'eps' in state_dict:
            self._eps_value = float(state_dict[prefix + 'eps'].reshape(-1)[0])

    def _take(self, g, inds, sl):
        return g[:, sl] if sl is not None else g[:, inds]

    def _scatter(self, like, values):
        sl = self._warp_sl
        if sl is not None and sl.step in (None, 1):
            # a contiguous block of warped indices: zero padding = one kernel forward, a narrow backward (an in-place slice
            # assignment costs zeros + copy forward and a clone + copy in its CopySlices backward)
            return torch.nn.functional.pad(values, (sl.start, like.shape[1] - sl.stop))
        out = torch.zeros_like(like)
        if self._warp_sl is not None:
            out[:, self._warp_sl] = values
        else:
            out[:, self.warp_inds] = values
        return out

    def forward(self, g, mode='direct'):
        kept = self._take(g, self.keep_inds, self._keep_sl).contiguous()
        logvar = self._scatter(g, torch.log(self.eps + torch.exp(self.T_logvar_0(kept))))
        mu = self._scatter(g, self.T_mu_0(kept))
        if mode == 'direct':
            g_out = torch.exp(0.5 * logvar) * g + mu
        elif mode == 'inverse':
            g_out = torch.exp(-0.5 * logvar) * (g - mu)
        else:
            raise ValueError(f"mode must be 'direct' or 'inverse', got {mode!r}")
        return g_out, mu, logvar


class RealNVPFlowCouple(nn.Module):
    """Two complementary couplings (reference flows.py:219-243): pattern 0 = even / odd, pattern 1 = first / second half."""

    def __init__(self, n_features, g_n_features, weight_std=0.01, pattern=0):
        super().__init__()
        self.n_features, self.g_n_features, self.weight_std, self.pattern = n_features, g_n_features, weight_std, pattern
        idx = list(range(g_n_features))
        if pattern == 0:
            w1, w2 = idx[::2], idx[1::2]
        elif pattern == 1:
            w1, w2 = idx[:g_n_features // 2], idx[g_n_features // 2:]
        else:
            raise ValueError('pattern must be 0 or 1')
        self.nvp1 = RealNVPFlow(n_features, g_n_features, weight_std=weight_std, warp_inds=w1)
        self.nvp2 = RealNVPFlow(n_features, g_n_features, weight_std=weight_std, warp_inds=w2)

    def forward(self, g, mode='direct'):
        if mode == 'direct':
            g1, mu1, lv1 = self.nvp1(g, mode=mode)
            g2, mu2, lv2 = self.nvp2(g1, mode=mode)
        elif mode == 'inverse':
            g2, mu2, lv2 = self.nvp2(g, mode=mode)
            g1, mu1, lv1 = self.nvp1(g2, mode=mode)
        else:
            raise ValueError(f"mode must be 'direct' or 'inverse', got {mode!r}")
        return [g1, g2], [mu1, mu2], [lv1, lv2]


class GlobalRNVPDecoder(nn.Module):
    """Stack of couples; lists are direct-ordered in both modes (reference decoders.py:7-38)."""

    def __init__(self, n_flows, n_features, g_n_features, weight_std=0.01):
        super().__init__()
        self.n_flows, self.n_features, self.g_n_features, self.weight_std = n_flows, n_features, g_n_features, weight_std
        self.flows = nn.ModuleList([RealNVPFlowCouple(n_features, g_n_features, weight_std=weight_std, pattern=i % 2)
                                    for i in range(n_flows)])

    # ---- fused HIP path ------------------------------------------------------------------------------------------------
    def _bn_modules(self):
        return [getattr(nvp, f'T_{X}_0')[1] for couple in self.flows for nvp in (couple.nvp1, couple.nvp2) for X in ('mu', 'logvar')]

    def _raw_arena(self):
        """Parameters + BatchNorm buffers in the record order of csrc/gwtf_prior.hip (one autograd-aware torch.cat)."""
        parts = []
        for couple in self.flows:
            for nvp in (couple.nvp1, couple.nvp2):
                for X in ('mu', 'logvar'):
                    t = getattr(nvp, f'T_{X}_0')
                    parts += [t[0].weight, t[1].weight, t[1].bias, t[1].running_mean, t[1].running_var, t[3].weight, t[3].bias]
        return torch.cat([q.reshape(-1) for q in parts])

    def _syncs(self):
        """Batch statistics over the rows of ALL ranks (the reference converts every BatchNorm to SyncBatchNorm,
        train_ae.py:152)."""
        from .dist import syncs_statistics
        return self.training and syncs_statistics(self._bn_modules())

    def _fused_ok(self, g, rows=None):
        """The one-launch kernels cover this call (csrc/gwtf_prior.hip make_plan: any number of rows, G <= 512, F <= 128, fp32 affine
        BatchNorm with a momentum); otherwise the module-by-module evaluation runs."""
        bns = self._bn_modules()
        rows = g.shape[0] if rows is None else rows
        return (g.is_cuda and g.dim() == 2 and 2 <= g.shape[1] == self.g_n_features <= 512 and rows >= 1 and
                self.n_features <= 128 and 2 * self.n_flows <= 64 and g.dtype == torch.float32 and
                all(bn.track_running_stats and bn.momentum is not None and bn.affine and bn.weight.dtype == torch.float32
                    for bn in bns))

    def _forward_fused(self, g, mode):
        if mode not in ('direct', 'inverse'):
            raise ValueError(f"mode must be 'direct' or 'inverse', got {mode!r}")
        training = self.training
        eps = self.flows[0].nvp1._eps_value
        row0, rows = 0, g.shape[0]
        if self._syncs():
            # data-parallel run: the flow is per SHAPE (a few MFLOP on one compute unit), so every rank evaluates it on the rows
            # of ALL ranks -- one all-gather forward, one all-reduce of the row gradients backward -- instead of synchronising
            # the statistics of each of its 4 n_flows BatchNorm layers; its own rows are sliced out below.  Same numbers as
            # SyncBatchNorm (each rank's loss sees every row through the batch statistics; GatherRows.backward sums that).
            from .dist import gather_rows
            g, lay = gather_rows(g.contiguous().float())
            row0 = lay.row0
        if training and g.shape[0] < 2:      # the (GLOBAL) batch BatchNorm sees: one local row is fine when other ranks hold more
            raise ValueError('Expected more than 1 value per channel when training (BatchNorm over the batch)')
        gs, mus, lvs, stats = _PriorFlowFn.apply(g.contiguous().float(), self._raw_arena(), self.n_flows, self.n_features,
                                                 eps, mode, training)
        if g.shape[0] != rows:
            gs, mus, lvs = (t[:, row0:row0 + rows] for t in (gs, mus, lvs))
        # the lists below are unbound views of these stacked tensors: a consumer that only needs sum_j logvars_j (GaussianFlowNLL) can
        # take lvs.sum(0) -- one kernel forward, one expand backward -- instead of 2 n_flows - 1 adds and an UnbindBackward per slot
        _STACKED[self] = (gs, mus, lvs)
        if training:
            # running = (1 - m) running + m batch, unbiased batch variance, num_batches_tracked += 1 (nn.BatchNorm1d)
            B = g.shape[0]
            mods = self._bn_modules()
            with torch.no_grad():
                flat = stats.reshape(len(mods), 2, self.n_features)
                means, variances = list(flat[:, 0].unbind(0)), list((flat[:, 1] * (B / (B - 1.0))).unbind(0))
                mom = [float(m.momentum) for m in mods]
                rms, rvs = [m.running_mean for m in mods], [m.running_var for m in mods]
                if len(set(mom)) == 1:
                    torch._foreach_mul_(rms, 1.0 - mom[0]); torch._foreach_add_(rms, means, alpha=mom[0])
                    torch._foreach_mul_(rvs, 1.0 - mom[0]); torch._foreach_add_(rvs, variances, alpha=mom[0])
                else:
                    for m, mu_, va, mo in zip(mods, means, variances, mom):
                        m.running_mean.mul_(1.0 - mo).add_(mu_, alpha=mo)
                        m.running_var.mul_(1.0 - mo).add_(va, alpha=mo)
                torch._foreach_add_([m.num_batches_tracked for m in mods], 1)
        return list(gs.unbind(0)), list(mus.unbind(0)), list(lvs.unbind(0))

    def forward_async(self, g, mode='direct'):
        """The fused launch on a SIDE stream: returns a handle whose ``result()`` joins the side stream into the current one
        and gives the usual (gs, mus, logvars) lists.  The whole prior flow is one workgroup on one compute unit (a latency
        chain, csrc/gwtf_prior.hip) and its outputs are only needed by the loss, so the caller can run the decoders -- which
        fill the other 255 compute units -- in between: the prior flow then costs no time on the critical path, forward or
        backward (autograd runs a node's backward on the stream of its forward).  Capturable: the fork / join pair becomes a
        branch of the hipGraph.  Falls back to the synchronous evaluation when the fused path does not apply."""
        _STACKED.pop(self, None)
        if not self._fused_ok(g, self._rows_seen(g)) or os.environ.get('GWTF_PRIOR_SIDE_STREAM', '1') == '0':
            return _PriorResult(self.forward(g, mode), None, None)
        cur = torch.cuda.current_stream(g.device)
        side = _SIDE_STREAMS.get(g.device)              # per device, not per module: a stream is not module state (deepcopy, pickle)
        if side is None:
            side = _SIDE_STREAMS[g.device] = torch.cuda.Stream(device=g.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            res = self._forward_fused(g, mode)
        g.record_stream(side)
        return _PriorResult(res, side, cur)

    def _rows_seen(self, g):
        """Rows the BatchNorm statistics cover: this rank's, or every rank's in a synchronised data-parallel run."""
        if g.is_cuda and self._syncs():
            from .dist import row_layout
            return row_layout(g.shape[0], g.device).total
        return g.shape[0]

    def forward(self, g, mode='direct'):
        _STACKED.pop(self, None)
        if self._fused_ok(g, self._rows_seen(g)):
            return self._forward_fused(g, mode)
        gs, mus, logvars = [], [], []
        cur = g
        order = self.flows if mode == 'direct' else reversed(self.flows)
        for flow in order:
            a, b, c = flow(cur, mode=mode)
            if mode == 'direct':
                gs, mus, logvars = gs + a, mus + b, logvars + c
                cur = gs[-1]
            else:
                gs, mus, logvars = a + gs, b + mus, c + logvars
                cur = gs[0]
        return gs, mus, logvars


_SIDE_STREAMS = {}
# module -> the stacked (gs, mus, logvars) of its LAST fused forward (kept outside the module: graph tensors in a module's __dict__
# would break copy.deepcopy / pickling of a model that has run)
import weakref
_STACKED = weakref.WeakKeyDictionary()


def stacked_lists(module):
    """ONE-SHOT: hands the stacked tensors of the module's last fused forward over and forgets them -- the entry must not keep
    an autograd graph alive beyond the step that made it (a hipGraph capture crashes when an earlier iteration's graph lives)."""
    return _STACKED.pop(module, None)


class _PriorResult:
    """Handle of GlobalRNVPDecoder.forward_async."""

    def __init__(self, res, side, origin):
        self._res, self._side, self._origin = res, side, origin

    def result(self):
        if self._side is not None:
            cur = torch.cuda.current_stream(self._side.device)
            cur.wait_stream(self._side)
            for lst in self._res:
                for t in lst:
                    t.record_stream(cur)
            self._side = None
        return self._res


class _PriorFlowFn(torch.autograd.Function):
    """gs, mus, logvars (2 n_flows, B, G) + BatchNorm batch statistics of the whole prior stack: one HIP launch forward,
    one backward (csrc/gwtf_prior.hip).  gs / logvars slots are differentiable; a gradient through a mus slot raises."""

    @staticmethod
    def forward(ctx, g, raw, n_flows, F, eps, mode, training):
        from . import _lib
        L = _lib.lib()
        B, G = g.shape
        n2 = 2 * n_flows
        raw = raw.contiguous()
        if raw.numel() != L.gwtf_prior_raw_floats(n_flows, G, F):
            raise _lib.GwtfError(f'prior raw arena has {raw.numel()} floats, expected {L.gwtf_prior_raw_floats(n_flows, G, F)}')
        dev = g.device
        lists = torch.empty(3, n2, B, G, device=dev, dtype=torch.float32)
        ws = torch.empty(L.gwtf_prior_workspace_floats(B, G, F), device=dev, dtype=torch.float32)
        stats = torch.zeros(n2, 2, 2, F, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(L.gwtf_prior_forward(_lib._ptr(g, 'g'), _lib._ptr(raw, 'raw'), lists[0].data_ptr(), lists[1].data_ptr(),
                                            lists[2].data_ptr(), ws.data_ptr(), stats.data_ptr(), n_flows, B, G, F, float(eps),
                                            _lib._MODES[mode], int(bool(training)), _lib._stream(g)))
        ctx.save_for_backward(g, raw, lists)
        ctx.meta = (n_flows, F, float(eps), mode, bool(training))
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)
        return lists[0], lists[1], lists[2], stats

    @staticmethod
    def backward(ctx, g_gs, g_mus, g_lvs, _gst):
        from . import _lib
        if g_mus is not None:
            raise NotImplementedError('a gradient reached a mus[j] list entry of the prior flow: the HIP backward differentiates '
                                      'through gs[j] and logvars[j] only (no reference consumer uses mus[j], losses.py:24-33)')
        g, raw, lists = ctx.saved_tensors
        n_flows, F, eps, mode, training = ctx.meta
        L = _lib.lib()
        B, G = g.shape
        dev = g.device
        g_gs = g_gs.contiguous().float() if g_gs is not None else None
        g_lvs = g_lvs.contiguous().float() if g_lvs is not None else None
        ws = torch.empty(L.gwtf_prior_workspace_floats(B, G, F), device=dev, dtype=torch.float32)
        g_raw = torch.zeros_like(raw)
        g_g = torch.empty_like(g)
        with torch.cuda.device(dev):
            _lib.check(L.gwtf_prior_backward(g.data_ptr(), raw.data_ptr(), lists[0].data_ptr(), lists[1].data_ptr(),
                                             lists[2].data_ptr(), g_gs.data_ptr() if g_gs is not None else None,
                                             g_lvs.data_ptr() if g_lvs is not None else None, ws.data_ptr(), g_raw.data_ptr(),
                                             g_g.data_ptr(), n_flows, B, G, F, eps, _lib._MODES[mode], int(training),
                                             _lib._stream(g)))
        return g_g, g_raw, None, None, None, None, None


class LatentLossFn(torch.autograd.Function):
    """(loss, pnll, gnll, gent) of Flow_Mixture_Loss (reference losses.py:159-170) from the per-shape point NLL and the prior's
    tensors: one HIP launch forward, one backward (csrc/gwtf_latent.hip) in place of ~55 elementwise / reduction launches."""

    @staticmethod
    def forward(ctx, nll, z, mu0, lv0, flow_lv, post_lv, pw, gw, ew):
        from . import _lib
        B, G = z.shape
        n2 = flow_lv.shape[0]
        if nll.shape != (B,) or mu0.shape != (G,) or lv0.shape != (G,) or flow_lv.shape != (n2, B, G) or post_lv.shape != (B, G):
            raise _lib.GwtfError(f'latent loss: shapes {tuple(nll.shape)} {tuple(z.shape)} {tuple(mu0.shape)} {tuple(lv0.shape)} '
                                 f'{tuple(flow_lv.shape)} {tuple(post_lv.shape)} are not (B,), (B,G), (G,), (G,), (n2,B,G), (B,G)')
        L, P = _lib.lib(), _lib._ptr
        ws = L.gwtf_latent_loss_workspace_floats(B, G)
        buf = torch.empty(ws + 4, device=z.device, dtype=torch.float32)
        out = buf[ws:]
        _lib.check(L.gwtf_latent_loss_forward(P(nll, 'nll'), P(z, 'z'), P(mu0, 'mu0'), P(lv0, 'lv0'), P(flow_lv, 'flow_lv'),
                                              P(post_lv, 'post_lv'), buf.data_ptr(), out.data_ptr(), B, G, n2, pw, gw, ew,
                                              _lib._stream(z)))
        ctx.save_for_backward(z, mu0, lv0)
        ctx.cfg = (B, G, n2, pw, gw, ew)
        return out

    @staticmethod
    def backward(ctx, g_out):
        from . import _lib
        z, mu0, lv0 = ctx.saved_tensors
        B, G, n2, pw, gw, ew = ctx.cfg
        g_out = g_out.contiguous()
        buf = torch.empty(B + (n2 + 2) * B * G + 2 * G, device=z.device, dtype=torch.float32)
        g_nll, g_z, g_post, g_flow, g_mu0, g_lv0 = buf.split([B, B * G, B * G, n2 * B * G, G, G])
        _lib.check(_lib.lib().gwtf_latent_loss_backward(_lib._ptr(g_out, 'g_out'), z.data_ptr(), mu0.data_ptr(), lv0.data_ptr(),
                                                        g_nll.data_ptr(), g_z.data_ptr(), g_mu0.data_ptr(), g_lv0.data_ptr(),
                                                        g_flow.data_ptr(), g_post.data_ptr(), B, G, n2, pw, gw, ew, _lib._stream(z)))
        return g_nll, g_z.view(B, G), g_mu0, g_lv0, g_flow.view(n2, B, G), g_post.view(B, G), None, None, None


class GaussianFlowNLL(nn.Module):
    """reference losses.py:24-33: 0.5 * (sum(sum_j logvars_j + (z - mu0)^2 / exp(logvar0)) / B + G log 2 pi)."""

    def forward(self, samples, mus, logvars, stacked_flow_logvars=None):
        """stacked_flow_logvars: logvars[1:] as ONE (2 n_flows, B, G) tensor when the fused prior flow produced them (same values;
        the sum then costs one kernel instead of 2 n_flows adds)."""
        z, B, G = samples[0], samples[0].shape[0], samples[0].shape[1]
        lv_sum = sum(logvars) if stacked_flow_logvars is None else logvars[0] + stacked_flow_logvars.sum(0)
        total = torch.sum(lv_sum + (z - mus[0]) ** 2 / torch.exp(logvars[0])) / B
        return 0.5 * (total + float(np.log(2.0 * np.pi)) * G)


class GaussianEntropy(nn.Module):
    """reference losses.py:36-41."""

    def forward(self, logvars):
        return 0.5 * (logvars.shape[1] * (1.0 + float(np.log(2.0 * np.pi))) + logvars.sum(1).mean())
