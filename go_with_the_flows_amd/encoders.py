"""Point-cloud encoder and the per-shape heads that sit immediately before the point-flow decoder.

Host-side mirror of lib/networks/encoders.py (PointNetCloudEncoder :9-28, FeatureEncoder :31-85, WeightsEncoder :87-91):
same constructors, attribute names and ``state_dict`` keys, so the reference's checkpoints load.

PointNetCloudEncoder in eval mode under ``torch.no_grad()`` runs the fused HIP kernel (csrc/gwtf_encoder.hip: all
SharedDot+BatchNorm+ReLU layers chained in registers on the MFMA, optional max-pool fused).  ``forward_max`` under
``.train()`` (batch-statistic BatchNorm, what the training step runs: models.py:127-128 inside training.py:43-54) is the
layer-at-a-time HIP pipeline of csrc/gwtf_encoder_train.hip, forward and backward (``_EncoderTrainFn``).  What is left to
plain library GEMMs and batch-norms on the HIP device (torch.matmul -> rocBLAS, F.batch_norm -> MIOpen, differentiated by
autograd): eval mode with a gradient required, ``forward`` (the full (B,C,N) feature map) in train mode, width lists
without kernels, an input that itself requires a gradient.  CPU tensors raise: there is no CPU path.  The per-shape heads
(FeatureEncoder, WeightsEncoder) run one HIP launch per layer, forward and backward (csrc/gwtf_heads.hip, ``_HeadLayerFn``), for any
number of rows (the gathered rows of a large data-parallel batch are walked 64 / 128 at a time inside the kernels).
"""
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib
from ._lib import GwtfError, _ptr, _stream, check
from .layers import SharedDot, Swish


def _bn_sync(bn):
    """True when this BatchNorm sums its batch statistics over the ranks (SyncBatchNorm after train_ae.py:152, more than
    one rank in the default group)."""
    from .dist import sharded
    return isinstance(bn, nn.SyncBatchNorm) and sharded()


class _EncoderTrainFn(torch.autograd.Function):
    """pooled (B,512) = max over points of the train-mode encoder, csrc/gwtf_encoder_train.hip.  params = (W, bn.weight,
    bn.bias) of the four layers; ``enc`` supplies the BatchNorm buffers (running statistics are updated in place, as
    F.batch_norm does)."""

    @staticmethod
    def forward(ctx, x, enc, sync, *params):
        import torch.distributed as dist
        from .dist import run as _run
        L = _lib.lib()
        x = x.contiguous()
        B, _, N = x.shape
        dev, st = x.device, _stream(x)
        R = _lib.STAT_REPLICAS
        C = enc._widths
        bns = [m for m in enc.features.children() if isinstance(m, nn.modules.batchnorm._BatchNorm)]
        Ws = [w.detach().reshape(C[l + 1], C[l]).contiguous() for l, w in enumerate(params[0::3])]
        gam = [g.detach().contiguous() for g in params[1::3]]
        bet = [b.detach().contiguous() for b in params[2::3]]
        # points the statistics cover: every rank's shapes (the per-rank batch may differ by one, train_ae.py:77-78; the counts
        # are exchanged once and cached, dist.row_layout)
        from .dist import row_layout
        n_total = float((row_layout(B, dev).total if sync else B) * N)
        f32 = dict(device=dev, dtype=torch.float32)
        P = lambda t: 0 if t is None else _ptr(t, 'encoder buffer')
        rm = [bn.running_mean if bn.track_running_stats else None for bn in bns]
        rv = [bn.running_var if bn.track_running_stats else None for bn in bns]

        def over_ranks(t):
            if sync:
                t = t.clone()
                _run(dist.all_reduce, t, op=dist.ReduceOp.SUM)
            return t

        def compact(slab):
            """sum of the R replicas of a statistic slab (R, ...) in one small launch (a torch sum over 64 rows takes 7 us)"""
            out = torch.empty(slab.shape[1:], **f32)
            check(L.gwtf_stat_compact(P(slab), P(out), slab.shape[0], out.numel(), st))
            return out

        with torch.cuda.device(dev):
            # every zero-initialised accumulator of the forward pass from ONE fill: {coordinate moments, layer sums, maxima, keys}
            sizes = [R * 12] + [R * 2 * C[l + 1] for l in (1, 2, 3)] + [4, 2 * (2 * B * C[4])]
            zero = torch.zeros(sum(sizes), **f32).split(sizes)
            mom, sums_l, ymax = zero[0].view(R, 12), {l: zero[l].view(R, 2, C[l + 1]) for l in (1, 2, 3)}, zero[4]
            keys = zero[5].view(torch.int64).view(2, B, C[4])                    # arg-max / arg-min keys of y_3 (never stored)
            check(L.gwtf_enc_train_xmoments(P(x), P(mom), B, N, st))
            mom_local = compact(mom)
            aff = [torch.empty(4 * C[l + 1], **f32) for l in range(4)]
            table0 = torch.empty(4 * C[1], **f32)
            check(L.gwtf_enc_train_fold0(P(over_ranks(mom_local)), n_total, P(Ws[0]), P(gam[0]), P(bet[0]), P(rm[0]), P(rv[0]),
                                         float(bns[0].momentum), P(aff[0]), P(table0), st))
            units_f, units_b = [None], [None]
            for l in (1, 2, 3):
                n_units = L.gwtf_enc_train_units_floats(l)
                units_f.append(torch.empty(n_units, **f32))
                units_b.append(torch.empty(n_units, **f32))
            check(L.gwtf_enc_train_pack_all(P(Ws[1]), P(Ws[2]), P(Ws[3]), P(units_f[1]), P(units_b[1]), P(units_f[2]), P(units_b[2]),
                                            P(units_f[3]), P(units_b[3]), st))       # one launch for the six fragment images
            ys = [None]
            for l in (1, 2, 3):
                # y_1, y_2 (and the backward's dA arrays) in tiles of 32 points x all channels: include/gwtf.h gwtf_enc_train_act_floats
                y = torch.empty(L.gwtf_enc_train_act_floats(B, C[l + 1], N), **f32) if l < 3 else None
                sums = sums_l[l]
                check(L.gwtf_enc_train_forward(l, P(x if l == 1 else ys[l - 1]), P(table0 if l == 1 else aff[l - 1]),
                                               P(units_f[l]), P(y), P(sums), P(ymax[l:l + 1]),
                                               keys[0].data_ptr() if l == 3 else 0, keys[1].data_ptr() if l == 3 else 0,
                                               P(gam[3]) if l == 3 else 0, B, N, st))
                check(L.gwtf_enc_train_fold(P(over_ranks(compact(sums))), l, n_total, P(gam[l]), P(bet[l]), P(rm[l]), P(rv[l]),
                                            float(bns[l].momentum), P(aff[l]), P(aff[l - 1]), st))
                ys.append(y)
            pooled = torch.empty(B, C[4], **f32)
            amax = torch.empty(B, C[4], device=dev, dtype=torch.int32)
            ystar = torch.empty(B, C[4], **f32)
            check(L.gwtf_enc_train_pool(keys[0].data_ptr(), keys[1].data_ptr(), P(aff[3]), P(pooled), amax.data_ptr(), P(ystar), B, N, st))
            counters = [bn.num_batches_tracked for bn in bns if bn.track_running_stats and bn.num_batches_tracked is not None]
            if counters:
                torch._foreach_add_(counters, 1)                                 # (one launch for the four counters)
        ctx.save_for_backward(x, pooled, *params)
        ctx.buf = dict(ys=ys, aff=aff, table0=table0, units_b=units_b, ymax=ymax, amax=amax, ystar=ystar,
                       mom_local=mom_local, Ws=Ws, gam=gam)
        ctx.meta = (B, N, n_total, sync, tuple(C))
        ctx.mark_non_differentiable(amax)
        return pooled, amax

    @staticmethod
    def backward(ctx, g_pooled, _g_amax):
        import torch.distributed as dist
        from .dist import run as _run
        L = _lib.lib()
        x, pooled, *params = ctx.saved_tensors
        b = ctx.buf
        B, N, n_total, sync, C = ctx.meta
        ys, aff, Ws, gam = b['ys'], b['aff'], b['Ws'], b['gam']
        dev, st = x.device, _stream(x)
        R = _lib.STAT_REPLICAS
        f32 = dict(device=dev, dtype=torch.float32)
        P = lambda t: 0 if t is None else _ptr(t, 'encoder buffer')

        def over_ranks(t):
            if sync:
                t = t.clone()
                _run(dist.all_reduce, t, op=dist.ReduceOp.SUM)
            return t

        def compact(slab):
            out = torch.empty(slab.shape[1:], **f32)
            check(L.gwtf_stat_compact(P(slab), P(out), slab.shape[0], out.numel(), st))
            return out

        grads = [None] * 12
        with torch.cuda.device(dev):
            g_pooled = g_pooled.contiguous().float()
            gp = torch.empty(B, C[4], **f32)
            red = torch.empty(2, C[4], **f32)
            # every zero-initialised accumulator of the backward pass from ONE fill
            sizes = [4, R * 3 * C[3], R * 2 * C[2], R * 5 * C[1]]
            zero = torch.zeros(sum(sizes), **f32).split(sizes)
            gmax, sums_l = zero[0], {3: zero[1].view(R, 3, C[3]), 2: zero[2].view(R, 2, C[2]), 1: zero[3].view(R, 5, C[1])}
            check(L.gwtf_enc_train_top(P(g_pooled), P(pooled), P(b['ystar']), P(aff[3]), P(gp), P(red), P(gmax[3:4]), B, st))
            partials = torch.empty(max(L.gwtf_enc_train_dw_partial_floats(l, B, N) for l in (1, 2, 3)), **f32)
            # ---- layer 3 in the M form (csrc/gwtf_encoder_train.hip): dy_3 = s gm_3 + Q y_3 + R with y_3 = W_3 a_2 ----
            grads[10], grads[11] = red[1], red[0]            # (views of `red`: nothing writes it again; a clone each was 8 launches per step)
            bconst = torch.empty(3 * C[4] + 4, **f32)
            check(L.gwtf_enc_train_bwd_consts(P(over_ranks(red[:2].contiguous())), 3, n_total, P(gam[3]), P(aff[3]), 0, 0, P(bconst), st))
            W3 = Ws[3]
            # M = W3^T diag(q3) W3 (256 x 256), its power-of-two operand scale, the fragment images of M 2^k and
            # mconst = {W3^T r3, 2^-k}: two launches (csrc/gwtf_encoder_glue.hip; as torch operators: a batched library GEMM, a
            # GEMV and 14 element-wise / reduction launches, ~85 us)
            units_m = torch.empty(L.gwtf_enc_train_units_floats(3) // 2, **f32)
            mconst = torch.empty(C[3] + 4, **f32)
            mws = torch.empty(L.gwtf_enc_train_mform_workspace_floats(C[3]), **f32)
            check(L.gwtf_enc_train_mform(P(W3), P(bconst), P(mws), P(units_m), P(mconst), C[3], C[4], st))
            extra = torch.empty(B, C[4], C[3], **f32)
            slot_of = torch.empty(B, N, device=dev, dtype=torch.int32)
            tables = torch.empty(B * (2 * C[4] + 2), device=dev, dtype=torch.int32)
            check(L.gwtf_enc_train_top_scatter(P(gp), bconst.data_ptr(), b['amax'].data_ptr(), P(W3), P(extra),      # coef = gp s_3
                                               slot_of.data_ptr(), tables.data_ptr(), B, N, st))
            up = torch.empty(L.gwtf_enc_train_act_floats(B, C[3], N), **f32)       # masked dL/da_2 (tiled like y_2)
            sums = sums_l[3]
            a2rows = torch.empty(B, C[4], C[3], **f32)                             # a_2 at the arg-max points, point-major
            check(L.gwtf_enc_train_backward_top(P(ys[2]), P(aff[2]), P(units_m), P(mconst), P(extra), slot_of.data_ptr(), P(up),
                                                P(sums), P(gmax[2:3]), P(a2rows), B, N, st))
            red = compact(sums)
            # dW_3 = s (.) S + Q (.) (W_3 G_2) + R (x) sum_p a_2  (gwtf_enc_train_dw3, gwtf_enc_train_dw3_finish)
            gram, S = torch.empty(C[3], C[3], **f32), torch.empty(C[4], C[3], **f32)
            check(L.gwtf_enc_train_dw3(P(gp), b['amax'].data_ptr(), slot_of.data_ptr(), P(a2rows), P(ys[2]), P(aff[2]), P(partials), P(gram),
                                       P(S), B, N, st))
            dW3 = torch.empty(C[4], C[3], **f32)
            check(L.gwtf_enc_train_dw3_finish(P(bconst), P(S), P(W3), P(gram), red[2].data_ptr(), P(dW3), C[3], C[4], st))
            grads[9] = dW3.view_as(params[9])
            for l in (2, 1):
                grads[3 * l + 1], grads[3 * l + 2] = red[1], red[0]                          # bn.weight, bn.bias of layer l
                bconst = torch.empty(3 * C[l + 1] + 4, **f32)
                check(L.gwtf_enc_train_bwd_consts(P(over_ranks(red[:2].contiguous())), l, n_total, P(gam[l]), P(aff[l]),
                                                  P(gmax[l:l + 1]), P(b['ymax'][l:l + 1]), P(bconst), st))
                dA = torch.empty(L.gwtf_enc_train_act_floats(B, C[l], N), **f32) if l > 1 else None
                sums = sums_l[l]
                check(L.gwtf_enc_train_backward(l, P(ys[l]), P(up), P(bconst), P(b['units_b'][l]), P(x if l == 1 else ys[l - 1]),
                                                P(aff[l - 1]), P(Ws[0] if l == 1 else None), P(dA), P(sums),
                                                P(gmax[l - 1:l]) if l > 1 else 0, B, N, st))
                dW = torch.empty(C[l + 1], C[l], **f32)
                check(L.gwtf_enc_train_dw(l, P(ys[l]), P(up), P(bconst), P(x if l == 1 else ys[l - 1]),
                                          P(b['table0'] if l == 1 else aff[l - 1]), P(partials), P(dW), B, N, st))
                grads[3 * l] = dW.view_as(params[3 * l])
                red = compact(sums)
                up = dA
            # layer 0 (3 -> 64): every sum its gradient needs is already there
            grads[1], grads[2] = red[1], red[0]
            bconst = torch.empty(3 * C[1] + 4, **f32)
            check(L.gwtf_enc_train_bwd_consts(P(over_ranks(red[:2].contiguous())), 0, n_total, P(gam[0]), P(aff[0]), 0, 0,
                                              P(bconst), st))
            dW0 = torch.empty(C[1], 3, **f32)
            check(L.gwtf_enc_train_dw0_finish(P(bconst), P(red), P(Ws[0]), P(b['mom_local']), P(dW0), C[1], st))
            grads[0] = dW0.view_as(params[0])
        return (None, None, None, *grads)


class PointNetCloudEncoder(nn.Module):
    def __init__(self, init_n_channels, init_n_features, n_features):
        super().__init__()
        self.init_n_channels = init_n_channels
        self.init_n_features = init_n_features
        self.n_features = n_features
        layers = OrderedDict()
        widths = [init_n_channels, init_n_features] + list(n_features)
        for i in range(1, len(widths)):
            name = 'init_sd' if i == 1 else f'sd{i - 2}'
            layers[name] = SharedDot(widths[i - 1], widths[i], 1, bias=False)
            layers[name + '_bn'] = nn.BatchNorm1d(widths[i])
            layers[name + '_relu'] = nn.ReLU(inplace=True)
        self.features = nn.Sequential(layers)
        self._widths = widths
        self._packed = None
        self._stamp = None

    # ---- packed weights (BatchNorm folded, split-f16 fragment order), cached per parameter version ----
    def _sources(self):
        out = []
        for name, mod in self.features.named_children():
            if isinstance(mod, SharedDot):
                out.append(mod.weight)
            elif isinstance(mod, nn.modules.batchnorm._BatchNorm):   # BatchNorm1d, or SyncBatchNorm after train_ae.py:152
                out += [mod.weight, mod.bias, mod.running_mean, mod.running_var]
        return out

    def invalidate_packed_weights(self):
        self._packed = None

    def train(self, mode=True):
        self._packed = None
        return super().train(mode)

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._packed = None
        return super().load_state_dict(*a, **k)

    def _widths_c(self):
        import ctypes
        return (ctypes.c_int * len(self._widths))(*self._widths), len(self._widths)

    def packed(self):
        src = self._sources()
        stamp = tuple((t.data_ptr(), t._version) for t in src)
        if self._packed is None or stamp != self._stamp:
            L = _lib.lib()
            w, n = self._widths_c()
            size = L.gwtf_encoder_packed_floats(w, n)
            if size == 0:
                raise GwtfError(f'no encoder kernel was built for widths {self._widths} '
                                '(built: [3,64,128,256,512] and [3,64,128,64,128]); see csrc/gwtf_encoder.hip')
            raw = torch.cat([t.detach().reshape(-1) for t in src])
            if raw.numel() != L.gwtf_encoder_raw_floats(w, n):
                raise GwtfError('encoder raw arena size mismatch')
            packed = torch.empty(size, device=raw.device, dtype=torch.float32)
            with torch.cuda.device(raw.device):
                check(L.gwtf_encoder_pack(_ptr(raw, 'raw'), _ptr(packed, 'packed'), w, n, _stream(raw)))
            self._packed, self._stamp = packed, stamp
        return self._packed

    def _fused(self, x, want_features, want_pooled):
        B, c, N = x.shape
        if c != 3 or self._widths[0] != 3:
            raise GwtfError(f'encoder input must be (B,3,N); got {tuple(x.shape)}')
        x = x.contiguous()
        packed = self.packed()
        C = self._widths[-1]
        feat = torch.empty(B, C, N, device=x.device, dtype=torch.float32) if want_features else None
        pooled = torch.empty(B, C, device=x.device, dtype=torch.float32) if want_pooled else None
        w, n = self._widths_c()
        with torch.cuda.device(x.device):
            check(_lib.lib().gwtf_encoder_forward(_ptr(x, 'input'), _ptr(packed, 'packed'), _ptr(feat, 'features'),
                                                  _ptr(pooled, 'pooled'), B, N, w, n, _stream(x)))
        return feat, pooled

    def has_fused_kernel(self):
        """True when csrc/gwtf_encoder.hip holds an instantiation for this width list."""
        w, n = self._widths_c()
        return _lib.lib().gwtf_encoder_packed_floats(w, n) != 0

    def _needs_graph(self, x):
        if self.training or (torch.is_grad_enabled() and
                             (x.requires_grad or any(p.requires_grad for p in self.parameters()))):
            return True
        return not self.has_fused_kernel()        # widths without an instantiation: the library-GEMM path on the device

    def forward(self, input):
        """(B,3,N) -> (B,C_last,N) per-point features (reference encoders.py:27-28)."""
        _ptr(input if input.is_contiguous() else input.contiguous(), 'input')      # device / dtype checks, loud on CPU
        if self._needs_graph(input):
            return self.features(input)
        return self._fused(input, True, False)[0]

    def _train_pipeline_ok(self, x):
        """The layer-at-a-time HIP train pipeline covers this call: batch-statistic BatchNorm with a momentum, the width
        list the kernels were built for, no gradient wanted for the points themselves, N a multiple of 4."""
        if not self.training or x.dim() != 3 or x.shape[1] != 3 or x.shape[2] % 4 or x.shape[0] == 0:
            return False
        if x.shape[2] > 12288:               # the arg-max row tables of the top layer's backward index the points of a shape in LDS
            return False
        if torch.is_grad_enabled() and x.requires_grad:
            return False
        bns = [m for m in self.features.children() if isinstance(m, nn.modules.batchnorm._BatchNorm)]
        if any(bn.momentum is None or not bn.affine or not bn.training for bn in bns):     # a frozen (eval) BatchNorm: library path
            return False
        if len({_bn_sync(bn) for bn in bns}) > 1:                                            # mixed plain / synchronised modules
            return False
        w, n = self._widths_c()
        return bool(_lib.lib().gwtf_enc_train_supported(w, n))

    def _train_params(self):
        out = []
        for mod in self.features.children():
            if isinstance(mod, SharedDot):
                out.append(mod.weight)
            elif isinstance(mod, nn.modules.batchnorm._BatchNorm):
                out += [mod.weight, mod.bias]
        return out

    def forward_max(self, input, return_indices=False):
        """(B,3,N) -> (B,C_last): features max-pooled over points, what models.py:127-128 consumes; eval/no-grad never
        materialises the (B,C_last,N) tensor, train mode runs csrc/gwtf_encoder_train.hip (forward and backward)."""
        _ptr(input if input.is_contiguous() else input.contiguous(), 'input')
        if self._train_pipeline_ok(input):
            bns = [m for m in self.features.children() if isinstance(m, nn.modules.batchnorm._BatchNorm)]
            pooled, amax = _EncoderTrainFn.apply(input, self, _bn_sync(bns[0]), *self._train_params())
            return (pooled, amax) if return_indices else pooled
        if self._needs_graph(input):
            pooled, amax = torch.max(self.features(input), dim=2)
            return (pooled, amax) if return_indices else pooled
        if return_indices:
            raise GwtfError('arg-max indices are not produced by the fused eval kernel')
        return self._fused(input, False, True)[1]


class _HeadLayerFn(torch.autograd.Function):
    """One layer of a per-shape head on the HIP device (csrc/gwtf_heads.hip): out = act(BatchNorm(x W^T + bias)), forward and
    backward one C call each.  bn: the BatchNorm module whose buffers the forward updates (None: no BatchNorm);
    bn_mode 0 none / 1 batch statistics / 2 running statistics; act 0 none / 1 swish / 2 log_softmax."""

    @staticmethod
    def forward(ctx, x, W, bias, gamma, beta, bn, bn_mode, bn_updates, act):
        L = _lib.lib()
        x, W = x.contiguous(), W.contiguous()
        B, Din = x.shape
        Dout = W.shape[0]
        dev = x.device
        ypre = torch.empty(B, Dout, device=dev, dtype=torch.float32)
        out = torch.empty(B, Dout, device=dev, dtype=torch.float32)
        stats = torch.empty(3, Dout, device=dev, dtype=torch.float32) if bn_mode else None
        P = lambda t: None if t is None else _ptr(t.contiguous() if not t.is_contiguous() else t, 'head operand')
        tracked = bn is not None and bn.track_running_stats and bn.running_mean is not None
        rm, rv = (bn.running_mean, bn.running_var) if tracked else (None, None)
        nbt = bn.num_batches_tracked.data_ptr() if (tracked and bn.num_batches_tracked is not None) else None
        with torch.cuda.device(dev):
            check(L.gwtf_head_layer_forward(P(x), P(W), P(bias), P(gamma), P(beta), P(rm), P(rv), nbt,
                                            float(bn.momentum) if bn is not None and bn.momentum is not None else 0.0,
                                            float(bn.eps) if bn is not None else 1e-5, bn_mode, int(bn_updates), act, P(ypre), P(stats),
                                            P(out), B, Din, Dout, _stream(x)))
        if bn_mode == 1 and tracked:
            torch._C._increment_version([t for t in (rm, rv, bn.num_batches_tracked) if t is not None])   # written through raw pointers
        ctx.save_for_backward(x, W, bias, gamma, beta, ypre, stats, out)
        ctx.meta = (bn_mode, act)
        return out

    @staticmethod
    def backward(ctx, g_out):
        L = _lib.lib()
        x, W, bias, gamma, beta, ypre, stats, out = ctx.saved_tensors
        bn_mode, act = ctx.meta
        B, Din = x.shape
        Dout = W.shape[0]
        dev = x.device
        new = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
        g_out = g_out.contiguous().float()
        need = ctx.needs_input_grad
        g_y = new(B, Dout)
        g_x = new(B, Din) if need[0] else None
        g_W = new(Dout, Din) if need[1] else None
        g_b = new(Dout) if (bias is not None and need[2]) else None
        g_ga = new(Dout) if (gamma is not None and need[3]) else None
        g_be = new(Dout) if (beta is not None and need[4]) else None
        P = lambda t: None if t is None else t.data_ptr()
        with torch.cuda.device(dev):
            check(L.gwtf_head_layer_backward(P(x), P(W), P(bias), P(gamma), P(beta), P(ypre), P(stats), P(out), P(g_out), bn_mode, act,
                                             P(g_y), P(g_x), 0, P(g_W), P(g_b), P(g_ga), P(g_be), B, Din, Dout, _stream(x)))
        return g_x, g_W, g_b, g_ga, g_be, None, None, None, None


class _HeadPairFn(torch.autograd.Function):
    """The mu and logvar heads of a FeatureEncoder (two plain Linear layers on the same hidden activations) in ONE launch forward and
    two backward (csrc/gwtf_heads.hip head_*_pair_kernel): the same arithmetic as two _HeadLayerFn calls."""

    @staticmethod
    def forward(ctx, x, Wa, ba, Wb, bb):
        L = _lib.lib()
        x, Wa, Wb = x.contiguous(), Wa.contiguous(), Wb.contiguous()
        B, Din = x.shape
        Da, Db = Wa.shape[0], Wb.shape[0]
        new = lambda *shape: torch.empty(*shape, device=x.device, dtype=torch.float32)
        ypre_a, out_a, ypre_b, out_b = new(B, Da), new(B, Da), new(B, Db), new(B, Db)
        P = lambda t: None if t is None else _ptr(t.contiguous() if not t.is_contiguous() else t, 'head operand')
        with torch.cuda.device(x.device):
            check(L.gwtf_head_pair_forward(P(x), P(Wa), P(ba), P(Wb), P(bb), P(ypre_a), P(out_a), P(ypre_b), P(out_b), B, Din, Da, Db,
                                           _stream(x)))
        ctx.save_for_backward(x, Wa, ba, Wb, bb, ypre_a, out_a, ypre_b, out_b)
        ctx.set_materialize_grads(False)       # an output the loss does not read arrives as None: its head's parameters get no gradient
        return out_a, out_b

    @staticmethod
    def backward(ctx, g_a, g_b):
        L = _lib.lib()
        x, Wa, ba, Wb, bb, ypre_a, out_a, ypre_b, out_b = ctx.saved_tensors
        B, Din = x.shape
        Da, Db = Wa.shape[0], Wb.shape[0]
        new = lambda *shape: torch.empty(*shape, device=x.device, dtype=torch.float32)
        if g_a is None and g_b is None:
            return None, None, None, None, None
        used_a, used_b = g_a is not None, g_b is not None
        g_a = torch.zeros_like(out_a) if g_a is None else g_a.contiguous().float()
        g_b = torch.zeros_like(out_b) if g_b is None else g_b.contiguous().float()
        need = ctx.needs_input_grad
        g_x = new(B, Din) if need[0] else None
        g_Wa, g_Wb = (new(Da, Din) if need[1] else None), (new(Db, Din) if need[3] else None)
        g_ba = new(Da) if (ba is not None and need[2]) else None
        g_bb = new(Db) if (bb is not None and need[4]) else None
        g_ya, g_yb = new(B, Da), new(B, Db)
        P = lambda t: None if t is None else t.data_ptr()
        with torch.cuda.device(x.device):
            check(L.gwtf_head_pair_backward(P(x), P(Wa), P(ba), P(Wb), P(bb), P(ypre_a), P(out_a), P(ypre_b), P(out_b), P(g_a), P(g_b),
                                            P(g_ya), P(g_yb), P(g_x), P(g_Wa), P(g_ba), P(g_Wb), P(g_bb), B, Din, Da, Db, _stream(x)))
        return g_x, (g_Wa if used_a else None), (g_ba if used_a else None), (g_Wb if used_b else None), (g_bb if used_b else None)


class FeatureEncoder(nn.Module):
    """Per-shape MLP with Gaussian heads (reference encoders.py:31-85)."""

    def __init__(self, n_layers, in_features, latent_space_size, deterministic=False, batch_norm=True,
                 mu_weight_std=0.001, mu_bias=0.0, logvar_weight_std=0.01, logvar_bias=0.0, easy_init=False):
        super().__init__()
        self.n_layers, self.in_features, self.latent_space_size = n_layers, in_features, latent_space_size
        self.deterministic, self.batch_norm = deterministic, batch_norm
        self.mu_weight_std, self.mu_bias = mu_weight_std, mu_bias
        self.logvar_weight_std, self.logvar_bias = logvar_weight_std, logvar_bias
        self.easy_init = easy_init
        if n_layers > 0:
            self.features = nn.Sequential()
            for i in range(n_layers):
                self.features.add_module(f'mlp{i}', nn.Linear(in_features, in_features, bias=False))
                if batch_norm:
                    self.features.add_module(f'mlp{i}_bn', nn.BatchNorm1d(in_features))
                self.features.add_module(f'mlp{i}_swish', Swish())
        self.mus = nn.Sequential(OrderedDict(mu_mlp0=nn.Linear(in_features, latent_space_size, bias=True)))
        if not easy_init:
            with torch.no_grad():
                self.mus[-1].weight.normal_(std=mu_weight_std)
                self.mus[-1].bias.fill_(mu_bias)
        if not deterministic:
            self.logvars = nn.Sequential(OrderedDict(logvar_mlp0=nn.Linear(in_features, latent_space_size, bias=True)))
            if not easy_init:
                with torch.no_grad():
                    self.logvars[-1].weight.normal_(std=logvar_weight_std)
                    self.logvars[-1].bias.fill_(logvar_bias)

    def _bn_modules(self):
        return [m for m in self.features if isinstance(m, nn.modules.batchnorm._BatchNorm)] if self.n_layers > 0 else []

    def _features_functional(self, h, bn_updates):
        """self.features(h) layer by layer with torch.nn.functional calls -- for the two cases the module call cannot serve:
        SyncBatchNorm modules whose input already holds the rows of all ranks (their own forward would synchronise again), and
        `bn_updates` > 1 = the running statistics advanced as if this batch had been seen that many times (the reference
        evaluates p_prior once per mixture component on the same latents, models.py:169-193 inside flow_mixture.py:163-166)."""
        for mod in self.features:
            if isinstance(mod, nn.Linear):
                h = nn.functional.linear(h, mod.weight, mod.bias)
            elif isinstance(mod, nn.modules.batchnorm._BatchNorm):
                batch_stats = mod.training or not mod.track_running_stats
                x = h
                h = nn.functional.batch_norm(x, mod.running_mean if mod.track_running_stats else None,
                                             mod.running_var if mod.track_running_stats else None, mod.weight, mod.bias,
                                             batch_stats, 0.0 if mod.momentum is None else mod.momentum, mod.eps)
                if mod.training and mod.track_running_stats:
                    if mod.momentum is None:
                        raise NotImplementedError('BatchNorm momentum=None (cumulative average) is not supported here; the '
                                                  'reference never sets it (encoders.py:49)')
                    with torch.no_grad():
                        if bn_updates > 1:
                            xd = x.detach()
                            keep = (1.0 - mod.momentum) ** (bn_updates - 1)
                            # .data: the batch-norm node saved these buffers for its backward (it only reads them in eval mode); the
                            # replayed updates must not trip autograd's version check -- K real passes update them in place too
                            mod.running_mean.data.mul_(keep).add_(xd.mean(0), alpha=1.0 - keep)
                            mod.running_var.data.mul_(keep).add_(xd.var(0, unbiased=True), alpha=1.0 - keep)
                        if mod.num_batches_tracked is not None:
                            mod.num_batches_tracked.data.add_(bn_updates)
            else:
                h = mod(h)
        return h

    def _hip_layers(self, input):
        """[(Linear, BatchNorm or None)] of the trunk when csrc/gwtf_heads.hip covers this call (fp32 rows on the HIP device -- any
        number of them: the kernels walk the batch in blocks --, plain Linear -> [BatchNorm1d with a momentum] -> Swish layers), else
        None: the library modules then run -- on the CPU (host-logic tests), for other dtypes."""
        if not (input.is_cuda and input.dim() == 2 and input.dtype == torch.float32 and input.shape[0] >= 1):
            return None
        layers, mods = [], list(self.features) if self.n_layers > 0 else []
        i = 0
        while i < len(mods):
            lin = mods[i]
            bn = mods[i + 1] if (i + 1 < len(mods) and isinstance(mods[i + 1], nn.modules.batchnorm._BatchNorm)) else None
            sw = mods[i + (2 if bn is not None else 1)] if i + (2 if bn is not None else 1) < len(mods) else None
            if not (isinstance(lin, nn.Linear) and lin.bias is None and isinstance(sw, Swish) and lin.weight.dtype == torch.float32):
                return None
            if bn is not None and (bn.momentum is None or (bn.training and input.shape[0] < 2)):
                return None
            layers.append((lin, bn))
            i += 3 if bn is not None else 2
        if any(max(l.weight.shape) > self.hip_max_width for l, _ in layers):
            return None
        ok = _lib.lib().gwtf_head_layer_supported
        dims = [(l.weight.shape[1], l.weight.shape[0], 1) for l, _ in layers] + \
               [(m[-1].weight.shape[1], m[-1].weight.shape[0], self._head_act) for m in ([self.mus] if self.deterministic else [self.mus, self.logvars])]
        if not all(ok(input.shape[0], din, dout, act) for din, dout, act in dims):
            return None
        return layers

    _head_act = 0          # activation code of the mu head in csrc/gwtf_heads.hip (WeightsEncoder: 2 = log_softmax)
    # Widest layer the HIP layer kernels take (a knob for A/B timing against the library modules: set it to 0 to get those).
    # Measured on the MI355X, forward + backward inside a hipGraph (tools/diag/heads_time.py, B = 64), HIP against library:
    # g_posterior (512 wide) 114 / 148 us, p_prior 80 / 110 (G = 128) and 102 / 116 (G = 512), the mixture-weight encoder
    # (3 trunk layers) 100 / 189 and 133 / 200.  (The first mapping -- 64 output columns per workgroup, K walked in dependent
    # passes -- lost at 512: 255 us; csrc/gwtf_gemm.h gemm_splitk_t has the reason and the fix.)
    hip_max_width = 4096

    def _hidden(self, input, bn_updates=1):
        """The shared trunk: Linear -> BatchNorm -> Swish per layer.  In a synchronised data-parallel run (SyncBatchNorm modules,
        train_ae.py:152) every rank evaluates the trunk on the rows of ALL ranks and keeps its own (dist.gather_rows): one
        all-gather forward and one all-reduce backward per module instead of two collectives per BatchNorm layer."""
        if self.n_layers == 0:
            return input
        bns = self._bn_modules()
        from .dist import gather_rows, syncs_statistics
        sync = self.training and bool(bns) and syncs_statistics(bns)
        rows = input.shape[0]
        if sync:
            input, lay = gather_rows(input)
        layers = self._hip_layers(input)
        if layers is not None:
            h = input
            for lin, bn in layers:
                mode = 0 if bn is None else (1 if (bn.training or not bn.track_running_stats) else 2)
                h = _HeadLayerFn.apply(h, lin.weight, None, bn.weight if bn is not None else None, bn.bias if bn is not None else None,
                                       bn, mode, bn_updates, 1)
        elif not sync and bn_updates == 1:
            h = self.features(input)
        else:
            h = self._features_functional(input, bn_updates)
        return h[lay.row0:lay.row0 + rows] if sync else h

    def _head(self, seq, h, act=0):
        lin = seq[-1]
        if h.is_cuda and h.dtype == torch.float32 and len(seq) == 1 and lin.weight.dtype == torch.float32 and \
                max(lin.weight.shape) <= self.hip_max_width and _lib.lib().gwtf_head_layer_supported(h.shape[0], lin.weight.shape[1], lin.weight.shape[0], act):
            return _HeadLayerFn.apply(h, lin.weight, lin.bias, None, None, None, 0, 1, act)
        out = seq(h)
        return nn.functional.log_softmax(out, dim=1) if act == 2 else out

    def forward(self, input, bn_updates=1):
        h = self._hidden(input, bn_updates)
        if self.deterministic:
            return self._head(self.mus, h, self._head_act)
        la, lb = self.mus[-1], self.logvars[-1]
        if (self._head_act == 0 and h.is_cuda and h.dtype == torch.float32 and len(self.mus) == 1 and len(self.logvars) == 1
                and la.weight.dtype == torch.float32 and lb.weight.dtype == torch.float32
                and max(*la.weight.shape, *lb.weight.shape) <= self.hip_max_width and os.environ.get('GWTF_NO_HEAD_PAIR') != '1'
                and _lib.lib().gwtf_head_layer_supported(h.shape[0], la.weight.shape[1], la.weight.shape[0], 0)
                and _lib.lib().gwtf_head_layer_supported(h.shape[0], lb.weight.shape[1], lb.weight.shape[0], 0)):
            return _HeadPairFn.apply(h, la.weight, la.bias, lb.weight, lb.bias)      # both heads: one launch
        return self._head(self.mus, h, self._head_act), self._head(self.logvars, h)


class WeightsEncoder(FeatureEncoder):
    """Mixture-weight head: log-softmax of the deterministic output (reference encoders.py:87-91)."""

    _head_act = 2          # the log_softmax is fused into the head's kernel (and applied by FeatureEncoder._head otherwise)
