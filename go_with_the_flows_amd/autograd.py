"""Differentiable density passes of a coupling stack: HIP forward + HIP backward behind torch.autograd.Function.

Structure (docs/LOG.md sections 4.6 / 4.7).  Everything per point -- the forward stack and, per coupling, the recompute,
dacc, dh = W1p^T dacc, the sd0 / FiLM-record gradient reductions, dx and the f x f weight gradient (accumulated inside
the backward kernel, summed by gwtf_dw1_reduce) -- is hand-written HIP (csrc/gwtf_stack.hip, gwtf_bwd.hip,
gwtf_train.hip).  What is O(f^2 + B*f*G) stays a small torch graph on views of ONE flat parameter arena: the "fold" from
module parameters and the latent g to the quantities the kernels consume (BatchNorm folded into sd0 / sd1, FiLM heads ->
per-shape {c, u}), so autograd carries the kernels' gradients to every reference parameter and to g.

  * eval BatchNorm (running statistics):  StackDensityFn                      -- one fused forward, one backward launch per coupling
  * train BatchNorm, any number of ranks: TrainMixtureFn                      -- the K-batched, phase-split C pipeline (gwtf_mtrain_*):
                                          all K mixture components per launch, ONE packed statistic all-reduce per phase
  * cross-check of the above:             MomentsFn / StatsFn / ApplyFn chain -- one autograd node per coupling (force_autograd_chain)

Reference semantics: loss.backward() through LocalCondRNVPDecoder.forward, training.py:54.
"""
import ctypes

import torch
import torch.nn.functional as F

from . import _lib

BN_EPS = 1e-5


def _w0_view(rec, pattern0, C, f):
    """sd0 weights of all couplings as (C,2,f,2).  The raw record holds sd0.weight as the module stores it: [f][2] for the
    couplings that keep two coordinates (natural view) and [f][1] followed by f zeros for those that keep one.  Built with
    slices only (no index tensors: the train step must stay hipGraph-capturable)."""
    if C % 3 == 0 and pattern0 % 3 == 0:            # decoder stacks: Triples alternate between the two kinds
        T, first_two = C // 3, (0 if pattern0 == 0 else 1)
        r = rec.reshape(T, 3, 2, 2 * f)
        out = rec.new_zeros(T, 3, 2, f, 2)
        out[first_two::2] = r[first_two::2].reshape(-1, 3, 2, f, 2)
        out[1 - first_two::2, :, :, :, 0] = r[1 - first_two::2][..., :f]
        return out.reshape(C, 2, f, 2)
    rows = []
    for c in range(C):                              # single couplings / odd stacks: a handful of records
        if (pattern0 + c) % 6 < 3:
            rows.append(rec[c].reshape(2, f, 2))
        else:
            rows.append(torch.stack([rec[c][:, :f], torch.zeros_like(rec[c][:, :f])], dim=-1))
    return torch.stack(rows)


def _gather(engine, raw=None):
    """The couplings' parameters / buffers as (C,2,...) tensors, branch order (logvar, mu): VIEWS of the flat raw arena
    (engine.raw_arena(): one autograd-aware torch.cat in the record order of csrc/gwtf_layout.h GwtfRaw), so gathering
    costs no kernels and every gradient flows back through that single cat."""
    if raw is None:
        raw = engine.raw_arena()
    C, f, G = engine.C, engine.f, engine.G
    rv = raw.view(C, 2, -1)                                   # (C, branch, branch record)
    o_bn0, o_w1, o_bn1 = 2 * f, 6 * f, 6 * f + f * f
    o_film, FS = 8 * f + f * f, f * G + 5 * f + f * f
    o_w2 = o_film + 2 * FS
    films = rv[:, :, o_film:o_w2].reshape(C, 2, 2, FS)         # (C, branch, {w,b}, head record)
    o_hbn, o_l1, o_b1 = f * G, f * G + 4 * f, f * G + 4 * f + f * f
    return {
        'raw': raw,
        'W0': _w0_view(rv[:, :, 0:2 * f], engine.pattern0, C, f),                    # (C,2,f,2)
        'bn0': list(rv[:, :, o_bn0:o_w1].reshape(C, 2, 4, f).unbind(2)),             # weight, bias, mean, var
        'W1': rv[:, :, o_w1:o_bn1].reshape(C, 2, f, f),
        'bn1': list(rv[:, :, o_bn1:o_film].reshape(C, 2, 2, f).unbind(2)),           # mean, var
        'W2': rv[:, :, o_w2:o_w2 + 2 * f].reshape(C, 2, 2, f),
        'b2': rv[:, :, o_w2 + 2 * f:o_w2 + 2 * f + 2],
        'L0': films[..., :o_hbn].reshape(C, 2, 2, f, G),
        'hbn': list(films[..., o_hbn:o_l1].reshape(C, 2, 2, 4, f).unbind(3)),
        'L1': films[..., o_l1:o_b1].reshape(C, 2, 2, f, f),
        'b1': films[..., o_b1:],
    }


def _gather_film(raw, C, f, G):
    """The FiLM-head slices of _gather for a (stacked) arena holding C couplings in all: what _film_train reads."""
    rv = raw.reshape(C, 2, -1)
    o_film, FS = 8 * f + f * f, f * G + 5 * f + f * f
    films = rv[:, :, o_film:o_film + 2 * FS].reshape(C, 2, 2, FS)
    o_hbn, o_l1, o_b1 = f * G, f * G + 4 * f, f * G + 4 * f + f * f
    return {'raw': raw, 'L0': films[..., :o_hbn].reshape(C, 2, 2, f, G),
            'hbn': list(films[..., o_hbn:o_l1].reshape(C, 2, 2, 4, f).unbind(3)),
            'L1': films[..., o_l1:o_b1].reshape(C, 2, 2, f, f), 'b1': films[..., o_b1:]}


def branch_poison(raw, C):
    """(C,2): 0 for a branch record whose parameters / buffers are all finite, NaN otherwise (0 * sum)."""
    return raw.detach().view(C, 2, -1).sum(-1) * 0.0


def fold(engine, g, eps):
    """Parameters + g -> folded tensors (all differentiable): W0f (C,2,f,2), c0f (C,2,f), W1p (C,2,f,f),
    cvec (B,C,2,f), u (B,C,2,2,f), b2 (C,2,2)."""
    P = _gather(engine)
    g0, be0, rm0, rv0 = P['bn0']
    s0 = g0 / torch.sqrt(rv0 + BN_EPS)
    W0f = P['W0'] * s0.unsqueeze(-1)
    c0f = be0 - rm0 * s0
    rm1, rv1 = P['bn1']
    s1 = 1.0 / torch.sqrt(rv1 + BN_EPS)
    W1p = P['W1'] * s1.unsqueeze(-1)
    c1 = -rm1 * s1
    # the FiLM heads (Linear -> BatchNorm with running statistics -> Swish -> Linear -> exp) in HIP, forward and backward
    # (csrc/gwtf_film_train.hip), parameters read in place from the arena; CPU / non-fp32 tensors raise (there is no torch
    # restatement in the package: tests/test_gpu_film_heads.py holds the one the kernels are checked against)
    film_raw, _, _ = FilmHeadsFn.apply(P['raw'], g, engine.C, engine.f, engine.G, 0, g.shape[0], eps, False)
    a, bsh = film_raw[:, :, :, 0, :engine.f], film_raw[:, :, :, 1, :engine.f]
    cvec = c1 + bsh / a
    u = P['W2'].unsqueeze(0) * a.unsqueeze(3)
    # Range scaling of the split-f16 contraction's operands, as the eval packer applies it (csrc/gwtf_layout.h, RANGE SCALING):
    # exact powers of two -- constants for autograd, which un-scales every gradient by the chain rule.
    with torch.no_grad():
        def floor_log2(t, lo, hi, shift=0):
            e = torch.frexp(t)[1].to(torch.float32) - 1.0 - shift
            ok = (t > 0) & torch.isfinite(t)
            return torch.where(ok, e.clamp(lo, hi), torch.zeros_like(e))
        cs = floor_log2(W0f.abs().sum(-1) + c0f.abs(), -40, 40)                       # (C,2,f)  per sd1 input feature
        rs = floor_log2((W1p * torch.exp2(cs).unsqueeze(-2)).abs().amax(-1), -50, 50, shift=12)   # (C,2,f)  per sd1 output row
        up_c, dn_c, up_r, dn_r = torch.exp2(cs), torch.exp2(-cs), torch.exp2(rs), torch.exp2(-rs)
    W0f, c0f = W0f * dn_c.unsqueeze(-1), c0f * dn_c
    W1p = W1p * (up_c.unsqueeze(-2) * dn_r.unsqueeze(-1))
    cvec = cvec * dn_r
    u = u * up_r.unsqueeze(0).unsqueeze(3)
    # a non-finite value anywhere in a branch record poisons that branch's sd2 biases (0 * sum = NaN), like the packer's POISON
    # slot: the kernels' v_max ReLU would otherwise turn e.g. a NaN sd0 weight into a zero activation
    return W0f, c0f, W1p, cvec, u, P['b2'] + branch_poison(P['raw'], W0f.shape[0]).unsqueeze(-1)


def film_record(cvec, u, b2, FP):
    """(B,C,2,f), (B,C,2,2,f), (C,2,2) -> the (B,C,6FP+4) record the kernels stage into LDS."""
    B, C, _, f = cvec.shape
    pad = (0, FP - f)
    parts = []
    for br in range(2):
        parts += [F.pad(cvec[:, :, br], pad), F.pad(u[:, :, br, 0], pad), F.pad(u[:, :, br, 1], pad)]
    parts.append(b2.reshape(1, C, 4).expand(B, C, 4))
    return torch.cat(parts, dim=-1).contiguous()


class StackDensityFn(torch.autograd.Function):
    """out, logdet, ps, mus, lvs = pass of the whole stack (either direction), differentiable w.r.t. p and the folded
    tensors through out, logdet AND every per-coupling list entry ps[j] / lvs[j] (the reference's list API returns
    differentiable tensors in every slot, decoders.py:61-79).  mus[j] is returned for the API; a gradient arriving through it
    raises (no reference consumer differentiates through mus)."""

    @staticmethod
    def forward(ctx, p, W0f, c0f, W1p, cvec, u, b2, C, f, pattern0, eps, mode):
        L = _lib.lib()
        FP = L.gwtf_padded_width(f)
        p = p.contiguous()
        dev = p.device
        pw = torch.empty(C * L.gwtf_packed_w_coupling_floats(f), device=dev, dtype=torch.float32)
        pb = torch.empty(C * L.gwtf_packed_b_coupling_floats(f), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(L.gwtf_pack_folded(W1p.contiguous().data_ptr(), W0f.contiguous().data_ptr(),
                                          c0f.contiguous().data_ptr(), pw.data_ptr(), pb.data_ptr(), C, f,
                                          _lib._stream(p)))
        rec = film_record(cvec, u, b2, FP)
        out, logdet, lists = _lib.stack_forward(p, pw, rec, C, f, pattern0, eps, mode, True)
        ctx.save_for_backward(p, lists[0], pw, pb, rec)
        ctx.meta = (C, f, FP, pattern0, eps, mode)
        ctx.set_materialize_grads(False)          # list slots nobody differentiates through arrive as None, not as zeros
        return out, logdet, lists[0], lists[1], lists[2]

    @staticmethod
    def backward(ctx, g_out, g_logdet, g_ps, g_mus, g_lvs):
        p, ps, pw, pb, rec = ctx.saved_tensors
        C, f, FP, pattern0, eps, mode = ctx.meta
        if g_mus is not None:
            raise NotImplementedError('a gradient reached a mus[j] list entry: the HIP backward differentiates through ps[j], '
                                      'logvars[j], the final coordinates and sum(logvars) only')
        inverse = mode == 'inverse'
        L = _lib.lib()
        B, _, N = p.shape
        dev = p.device
        R = _lib.STAT_REPLICAS
        PW, PB = L.gwtf_packed_w_coupling_floats(f), L.gwtf_packed_b_coupling_floats(f)
        cur = (g_out if g_out is not None else torch.zeros_like(p)).contiguous().float()
        g_ld = (g_logdet if g_logdet is not None else torch.zeros_like(p)).contiguous().float()
        g_ps = g_ps.contiguous().float() if g_ps is not None else None
        g_lvs = g_lvs.contiguous().float() if g_lvs is not None else None
        g_film = torch.zeros(B, C, 2, 3, FP, device=dev, dtype=torch.float32)
        g_sd0 = torch.zeros(C, R, 2, 3, FP, device=dev, dtype=torch.float32)
        g_bias = torch.zeros(C, R, 4, device=dev, dtype=torch.float32)
        ws = _lib.dw1_workspace(f, B, N, dev)            # per-workgroup dW1 partials of one backward pass
        gW1p = torch.empty(C, 2, f, f, device=dev, dtype=torch.float32)
        bufs = [torch.empty_like(p), torch.empty_like(p)]
        st = _lib._stream(p)
        with torch.cuda.device(dev):
            for c in (range(C) if inverse else range(C - 1, -1, -1)):   # reverse of the forward's processing order
                if inverse:
                    x_in = ps[c + 1] if c + 1 < C else p
                else:
                    x_in = ps[c - 1] if c > 0 else p
                nxt = bufs[c & 1]
                # gradients entering through this coupling's own list slots are added inside the kernel
                _lib.check(L.gwtf_coupling_backward_lists(x_in.data_ptr(), cur.data_ptr(), g_ld.data_ptr(),
                                                    g_ps[c].data_ptr() if g_ps is not None else None,
                                                    g_lvs[c].data_ptr() if g_lvs is not None else None,
                                                    pw[c * PW:].data_ptr(), pb[c * PB:].data_ptr(), rec.data_ptr(),
                                                    nxt.data_ptr(), ws.data_ptr(), g_film.data_ptr(),
                                                    g_sd0[c].data_ptr(), g_bias[c].data_ptr(), c, B, N, C, f, pattern0,
                                                    float(eps), _lib._MODES[mode], st))
                # dW1p[k][j][i] = sum_{b,n} dacc[k,j,(b,n)] h[k,i,(b,n)]: accumulated inside the kernel, summed here
                _lib.dw1_reduce(ws, 1, f, B, N, out=gW1p[c])
                cur = nxt
        gs = g_sd0.sum(1)                                              # (C,2,3,FP)
        g_W0f = gs[:, :, 0:2, :f].permute(0, 1, 3, 2).contiguous()    # (C,2,f,2)
        g_c0f = gs[:, :, 2, :f].contiguous()
        g_cvec = g_film[:, :, :, 0, :f].contiguous()
        g_u = g_film[:, :, :, 1:3, :f].contiguous()
        g_b2 = g_bias.sum(1).reshape(C, 2, 2)
        return cur, g_W0f, g_c0f, gW1p, g_cvec, g_u, g_b2, None, None, None, None, None


def density_forward(engine, p, g, mode='inverse'):
    """Differentiable (out, logdet, (ps, mus, lvs) stacked (C,B,3,N) each) of the whole stack in either direction;
    eval-mode BatchNorm."""
    eps = engine.couplings[0]._eps_value
    W0f, c0f, W1p, cvec, u, b2 = fold(engine, g.float(), eps)
    out, logdet, ps, mus, lvs = StackDensityFn.apply(p.float(), W0f, c0f, W1p, cvec, u, b2, engine.C, engine.f,
                                                     engine.pattern0, eps, mode)
    return out, logdet, (ps, mus, lvs)


# ======================================================================================================================
# Train mode: batch-statistic BatchNorm.  The statistics make every coupling depend on global reductions of its own
# input, so the differentiable path is a chain of small autograd nodes per coupling (reference flows.py:27,30,62,65 under
# model.train() + loss.backward()):
#     x --MomentsFn(HIP)--> M --fold0(torch)--> W0f,c0f --StatsFn(HIP)--> S --fold1(torch)--> c,u --ApplyFn(HIP)--> x', lv
# Autograd sums the three contributions to dL/dx (apply, statistics of y1, moments of x) and the two contributions to the
# sd0/sd1 weights; each HIP node's backward is one launch of csrc/gwtf_bwd.hip (+ the dW1 GEMM).
# ======================================================================================================================
def _pack_single(W1, W0f, c0f, f):
    L = _lib.lib()
    dev = W1.device
    pw = torch.empty(L.gwtf_packed_w_coupling_floats(f), device=dev, dtype=torch.float32)
    pb = torch.empty(L.gwtf_packed_b_coupling_floats(f), device=dev, dtype=torch.float32)
    w1, w0, c0 = W1.detach().contiguous(), W0f.detach().contiguous(), c0f.detach().contiguous()
    with torch.cuda.device(dev):
        _lib.check(L.gwtf_pack_folded(w1.data_ptr(), w0.data_ptr(), c0.data_ptr(), pw.data_ptr(), pb.data_ptr(), 1, f,
                                      torch.cuda.current_stream(dev).cuda_stream))
    return pw, pb


class MomentsFn(torch.autograd.Function):
    """x (B,3,N) -> the 9 first/second moments summed over all points {Sx0..2, Sx0x0, Sx0x1, Sx0x2, Sx1x1, Sx1x2, Sx2x2}."""

    @staticmethod
    def forward(ctx, x):
        L = _lib.lib()
        x = x.contiguous()
        B, _, N = x.shape
        mom = torch.zeros(_lib.STAT_REPLICAS, 16, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _lib.check(L.gwtf_train_moments(x.data_ptr(), mom.data_ptr(), B, N, _lib._stream(x)))
        ctx.save_for_backward(x)
        return mom.sum(0)[:9]

    @staticmethod
    def backward(ctx, gM):
        (x,) = ctx.saved_tensors
        Q = x.new_zeros(3, 3)
        idx = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
        for k, (a, b) in enumerate(idx):
            if a == b:
                Q[a, a] = 2.0 * gM[3 + k]
            else:
                Q[a, b] = gM[3 + k]
                Q[b, a] = gM[3 + k]
        return gM[:3].view(1, 3, 1) + torch.einsum('ab,zbn->zan', Q, x)


class StatsFn(torch.autograd.Function):
    """S[branch][feature][{sum y1, sum y1^2}] over all points, y1 = sd1(relu(W0f x_keep + c0f))  (csrc: stats_kernel)."""

    @staticmethod
    def forward(ctx, x, W0f, c0f, W1, pw, pb, pat, f):
        L = _lib.lib()
        x = x.contiguous()
        B, _, N = x.shape
        FP = L.gwtf_padded_width(f)
        ys = torch.zeros(_lib.STAT_REPLICAS, 2, FP, 2, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _lib.check(L.gwtf_train_stats(x.data_ptr(), pw.data_ptr(), ys.data_ptr(), B, N, f, pat, _lib.tune_word(), _lib._stream(x)))
        ctx.save_for_backward(x, pw, pb)
        ctx.meta = (pat, f, FP)
        return ys.sum(0)[:, :f, :]

    @staticmethod
    def backward(ctx, gS):
        x, pw, pb = ctx.saved_tensors
        pat, f, FP = ctx.meta
        L = _lib.lib()
        B, _, N = x.shape
        dev = x.device
        gst = torch.zeros(2, 2, FP, device=dev, dtype=torch.float32)
        gst[:, 0, :f] = gS[:, :, 0]
        gst[:, 1, :f] = gS[:, :, 1]
        g_x = torch.empty_like(x)
        ws = _lib.dw1_workspace(f, B, N, dev)
        g_sd0 = torch.zeros(_lib.STAT_REPLICAS, 2, 3, FP, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(L.gwtf_stats_backward(x.data_ptr(), gst.data_ptr(), pw.data_ptr(), pb.data_ptr(), g_x.data_ptr(),
                                             ws.data_ptr(), g_sd0.data_ptr(), B, N, f, pat, _lib._stream(x)))
        gW1 = _lib.dw1_reduce(ws, 1, f, B, N)
        gs = g_sd0.sum(0)
        return g_x, gs[:, 0:2, :f].permute(0, 2, 1).contiguous(), gs[:, 2, :f].contiguous(), gW1, None, None, None, None


class ApplyFn(torch.autograd.Function):
    """One coupling with given folded parameters: x -> (x_out, logvar, mu[detached])."""

    @staticmethod
    def forward(ctx, x, W0f, c0f, W1, cvec, u, b2, pw, pb, pat, f, eps, mode):
        L = _lib.lib()
        x = x.contiguous()
        FP = L.gwtf_padded_width(f)
        rec = film_record(cvec.unsqueeze(1), u.unsqueeze(1), b2.unsqueeze(0), FP)
        out, lv, lists = _lib.stack_forward(x, pw, rec, 1, f, pat, eps, mode, True)
        ctx.save_for_backward(x, pw, pb, rec)
        ctx.meta = (pat, f, FP, eps, mode)
        ctx.mark_non_differentiable(lists[1][0])
        return out, lv, lists[1][0]

    @staticmethod
    def backward(ctx, g_out, g_lv, _g_mu):
        x, pw, pb, rec = ctx.saved_tensors
        pat, f, FP, eps, mode = ctx.meta
        L = _lib.lib()
        B, _, N = x.shape
        dev = x.device
        R = _lib.STAT_REPLICAS
        g_out = (g_out if g_out is not None else torch.zeros_like(x)).contiguous()
        g_lv = (g_lv if g_lv is not None else torch.zeros_like(x)).contiguous()
        g_x = torch.empty_like(x)
        g_film = torch.zeros(B, 1, 2, 3, FP, device=dev, dtype=torch.float32)
        g_sd0 = torch.zeros(R, 2, 3, FP, device=dev, dtype=torch.float32)
        g_bias = torch.zeros(R, 4, device=dev, dtype=torch.float32)
        ws = _lib.dw1_workspace(f, B, N, dev)
        with torch.cuda.device(dev):
            _lib.check(L.gwtf_coupling_backward(x.data_ptr(), g_out.data_ptr(), g_lv.data_ptr(), pw.data_ptr(), pb.data_ptr(),
                                                rec.data_ptr(), g_x.data_ptr(), ws.data_ptr(),
                                                g_film.data_ptr(), g_sd0.data_ptr(), g_bias.data_ptr(), 0, B, N, 1, f, pat,
                                                float(eps), _lib._MODES[mode], _lib._stream(x)))
        gW1 = _lib.dw1_reduce(ws, 1, f, B, N)
        gs = g_sd0.sum(0)
        return (g_x, gs[:, 0:2, :f].permute(0, 2, 1).contiguous(), gs[:, 2, :f].contiguous(), gW1,
                g_film[:, 0, :, 0, :f].contiguous(), g_film[:, 0, :, 1:3, :f].contiguous(), g_bias.sum(0).reshape(2, 2),
                None, None, None, None, None, None)


class _BNSwishRows(torch.autograd.Function):
    """hraw (B,C,2,2,f), BatchNorm weight / bias (C,2,2,f) (strided views of the raw arena) -> swish(BatchNorm over the B rows with
    batch statistics), batch mean, biased batch variance: one HIP kernel per direction (csrc/gwtf_film.hip)."""

    @staticmethod
    def forward(ctx, hraw, hg, hb):
        L = _lib.lib()
        hraw = hraw.contiguous()
        B, C, f = hraw.shape[0], hraw.shape[1], hraw.shape[-1]
        M = hraw[0].numel()
        if hg.stride() != hb.stride() or hg.stride(3) != 1 or hg.dtype != torch.float32:
            hg, hb = hg.contiguous(), hb.contiguous()
        y = torch.empty_like(hraw)
        stats = torch.empty(3, M, device=hraw.device, dtype=torch.float32)
        with torch.cuda.device(hraw.device):
            _lib.check(L.gwtf_film_bn_swish_forward(hraw.data_ptr(), hg.data_ptr(), hb.data_ptr(), hg.stride(0), hg.stride(1),
                                                    hg.stride(2), f, B, M, y.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(),
                                                    stats[2].data_ptr(), _lib._stream(hraw)))
        ctx.save_for_backward(hraw, hg, hb, stats)
        mean, var = stats[0].view(hraw.shape[1:]), stats[1].view(hraw.shape[1:])
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    def backward(ctx, gy, _gm, _gv):
        hraw, hg, hb, stats = ctx.saved_tensors
        L = _lib.lib()
        B, f = hraw.shape[0], hraw.shape[-1]
        M = hraw[0].numel()
        gy = gy.contiguous()
        gx = torch.empty_like(hraw)
        gp = torch.empty(2, M, device=hraw.device, dtype=torch.float32)
        with torch.cuda.device(hraw.device):
            _lib.check(L.gwtf_film_bn_swish_backward(hraw.data_ptr(), gy.data_ptr(), hg.data_ptr(), hb.data_ptr(), hg.stride(0),
                                                     hg.stride(1), hg.stride(2), f, B, M, stats[0].data_ptr(), stats[2].data_ptr(),
                                                     gx.data_ptr(), gp[0].data_ptr(), gp[1].data_ptr(), _lib._stream(hraw)))
        return gx, gp[0].view(hraw.shape[1:]), gp[1].view(hraw.shape[1:])


PATHS = {'film_heads_hip': 0, 'film_heads_torch': 0}     # which FiLM-head implementation ran (tools/bench_train.py reports it)


def _film_train(P, g, eps):
    """FiLM heads with batch statistics over the B latent rows -> a, bsh (B,C,2,f) and batch {mean, unbiased var}.
    torch ops: only the per-coupling cross-check chain (force_autograd_chain) comes through here."""
    PATHS['film_heads_torch'] += 1
    hg, hb, _, _ = P['hbn']
    hraw = torch.einsum('bg,cxhfg->bcxhf', g, P['L0'])
    Bn = hraw.shape[0]
    hn, mean, var = _BNSwishRows.apply(hraw, hg, hb)
    o = torch.einsum('bcxhi,cxhji->bcxhj', hn, P['L1']) + P['b1']
    # + poison: a non-finite parameter anywhere in the branch makes the FiLM scale NaN, hence u = W2 a s1 and every output
    # (the kernels' v_max ReLU alone would turn e.g. a NaN sd0 weight into a zero activation)
    a = eps + torch.exp(o[:, :, :, 0]) + branch_poison(P['raw'], hraw.shape[1]).unsqueeze(0).unsqueeze(-1)
    return a, o[:, :, :, 1], mean.detach(), (var * (Bn / max(Bn - 1.0, 1.0))).detach()


class FilmHeadsFn(torch.autograd.Function):
    """The FiLM heads of KC couplings (all K stacks of a mixture at once) in HIP, forward and backward (csrc/gwtf_film_train.hip;
    reference flows.py:33-45, 68-80, 100-106).  raw: the (stacked) raw arena holding KC coupling records; g_all (B_all, G): the latent
    rows BatchNorm sees (all ranks' rows when data parallel); rows [row0, row0 + B) are this rank's shapes.
    -> film_raw (B, KC, 2, 2, FP) = {a = eps + exp(scale head), b = shift head} per branch, zero beyond the f valid columns (the
    train pipeline's record); batch mean / biased variance of the heads' BatchNorm (KC, 2, 2, f) (training) or its running ones."""

    @staticmethod
    def forward(ctx, raw, g_all, KC, f, G, row0, B, eps, training, handover=None):
        """handover: a dict shared with the TrainMixtureFn that consumes film_raw.  Both nodes produce a gradient for the SAME raw arena
        in disjoint slots (the pipeline: sd0 / sd1 / sd2 / BatchNorm records; this node: the FiLM records, written in place): the
        pipeline's backward -- which always runs first, film_raw being its input -- leaves its buffer in the dict and returns no raw
        gradient, and this node's backward writes its slots into THAT buffer and returns it: one gradient tensor instead of two 15-MB
        ones, no zero fill, no add (airplane config)."""
        ctx.handover, ctx.n_in = handover, (10 if handover is not None else 9)
        L = _lib.lib()
        raw, g_all = raw.contiguous(), g_all.contiguous().float()
        _lib._ptr(raw, 'the raw parameter arena')          # fp32, on a HIP device: a .double()'d or CPU module raises here
        _lib._ptr(g_all, 'g')
        if raw.device != g_all.device:
            raise _lib.GwtfError(f'FilmHeadsFn: parameters on {raw.device}, latents on {g_all.device}')
        dev = raw.device
        Ball, H, FP = g_all.shape[0], 4 * KC, L.gwtf_padded_width(f)
        if raw.numel() != KC * L.gwtf_raw_coupling_floats(f, G):
            raise _lib.GwtfError(f'FilmHeadsFn: arena of {raw.numel()} floats for {KC} couplings')
        PATHS['film_heads_hip'] += 1
        poison = branch_poison(raw, KC).contiguous()
        hraw = torch.empty(Ball, H, f, device=dev, dtype=torch.float32)
        hn = torch.empty_like(hraw)
        stats = torch.empty(3, H, f, device=dev, dtype=torch.float32)
        film_raw = torch.zeros(B, KC, 2, 2, FP, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(L.gwtf_film_heads_forward(raw.data_ptr(), g_all.data_ptr(), poison.data_ptr(), hraw.data_ptr(), hn.data_ptr(),
                                                 stats.data_ptr(), film_raw.data_ptr(), KC, f, G, Ball, row0, B, float(eps),
                                                 1 if training else 0, _lib._stream(raw)))
        ctx.save_for_backward(raw, g_all, hraw, hn, stats, film_raw)
        ctx.meta = (KC, f, G, row0, B, float(eps), bool(training))
        mean, var = stats[0].view(KC, 2, 2, f), stats[1].view(KC, 2, 2, f)
        ctx.mark_non_differentiable(mean, var)
        return film_raw, mean, var

    @staticmethod
    def backward(ctx, g_film_raw, _gm, _gv):
        raw, g_all, hraw, hn, stats, film_raw = ctx.saved_tensors
        KC, f, G, row0, B, eps, training = ctx.meta
        L = _lib.lib()
        dev = raw.device
        Ball = g_all.shape[0]
        g_film_raw = g_film_raw.contiguous().float()
        g_raw = ctx.handover.pop('g_raw', None) if ctx.handover is not None else None
        if g_raw is None or g_raw.shape != raw.shape:
            g_raw = torch.zeros_like(raw)
        dhraw = torch.empty_like(hraw)
        dg_part = torch.empty(L.gwtf_film_heads_slices(KC, G), Ball, G, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(L.gwtf_film_heads_backward(raw.data_ptr(), g_all.data_ptr(), hraw.data_ptr(), hn.data_ptr(), stats.data_ptr(),
                                                  film_raw.data_ptr(), g_film_raw.data_ptr(), g_raw.data_ptr(), dhraw.data_ptr(),
                                                  dg_part.data_ptr(), KC, f, G, Ball, row0, B, eps, 1 if training else 0,
                                                  _lib._stream(raw)))
        return (g_raw, dg_part.sum(0)) + (None,) * (ctx.n_in - 2)


class _AllReduceSum(torch.autograd.Function):
    """Sum over the ranks of the data-parallel group, in both directions: forward all-reduces the statistic,
    backward all-reduces its gradient (every rank's loss depends on every rank's points through the batch
    statistics -- SyncBatchNorm semantics, reference train_ae.py:152)."""

    @staticmethod
    def forward(ctx, t):
        import torch.distributed as dist
        from .dist import run
        t = t.clone()
        run(dist.all_reduce, t, op=dist.ReduceOp.SUM)
        return t

    @staticmethod
    def backward(ctx, g):
        import torch.distributed as dist
        from .dist import run
        g = g.clone().contiguous()
        run(dist.all_reduce, g, op=dist.ReduceOp.SUM)
        return g


def train_density_forward(engine, p, g, distributed=False, mode='inverse'):
    """Differentiable train-mode density pass.  Returns out, logdet, per-coupling lists (ps, mus, lvs in direct
    order; ps/lvs differentiable) and bn_batch (C,2,4,2,f) for the running-statistic update.
    distributed=True: statistics (and their gradients) are summed over torch.distributed's default group and the
    per-shape FiLM BatchNorm sees the latents of all ranks."""
    C, f, eps = engine.C, engine.f, engine.couplings[0]._eps_value
    B, _, N = p.shape
    P = _gather(engine)
    group_sum = _AllReduceSum.apply if distributed else None
    n = float(B * N)
    g = g.float()
    if distributed:
        from .dist import gather_rows
        g_all, lay = gather_rows(g)
        row0, n = lay.row0, float(lay.total * N)
        if g_all.shape[0] < 2:
            raise ValueError('train-mode BatchNorm needs more than 1 shape per (global) batch')
        a, bsh, fmean, fvar = _film_train(P, g_all, eps)
        a, bsh = a[row0:row0 + B], bsh[row0:row0 + B]
    else:
        if B < 2:
            raise ValueError('train-mode BatchNorm needs more than 1 shape per batch (torch raises the same)')
        a, bsh, fmean, fvar = _film_train(P, g, eps)
    bn_batch = torch.zeros(C, 2, 4, 2, f, device=p.device, dtype=torch.float32)
    bn_batch[:, :, 2:4, 0] = fmean
    bn_batch[:, :, 2:4, 1] = fvar
    g0, be0, _, _ = P['bn0']
    x = p.float().contiguous()
    ps, mus, lvs = [None] * C, [None] * C, [None] * C
    logdet = None
    for c in (range(C - 1, -1, -1) if mode == 'inverse' else range(C)):
        pat = (engine.pattern0 + c) % 6
        k0, k1 = {0: (1, 2), 1: (0, 2), 2: (0, 1), 3: (2, -1), 4: (1, -1), 5: (0, -1)}[pat]
        M = MomentsFn.apply(x)
        if group_sum is not None:
            M = group_sum(M)
        # fold0: sd0_bn statistics are analytic in the moments of the kept coordinates (double: E[xx]-E[x]E[x] cancels)
        Md = M.double()
        E = Md[:3] / n
        Sxx = torch.stack([torch.stack([Md[3], Md[4], Md[5]]), torch.stack([Md[4], Md[6], Md[7]]),
                           torch.stack([Md[5], Md[7], Md[8]])]) / n
        Cov = Sxx - torch.outer(E, E)
        zero = Md.new_zeros(())            # python-int indexing only: no index tensors (hipGraph-capturable)
        if k1 < 0:
            Ek = torch.stack([E[k0], zero])
            Ck = torch.stack([torch.stack([Cov[k0, k0], zero]), torch.stack([zero, zero])])
        else:
            Ek = torch.stack([E[k0], E[k1]])
            Ck = torch.stack([torch.stack([Cov[k0, k0], Cov[k0, k1]]), torch.stack([Cov[k1, k0], Cov[k1, k1]])])
        W0 = P['W0'][c].double()                                   # (2,f,2)
        mean0 = W0 @ Ek
        var0 = torch.einsum('xfa,ab,xfb->xf', W0, Ck, W0).clamp_min(0.0)
        s0 = (g0[c].double() / torch.sqrt(var0 + BN_EPS))
        W0f = (W0 * s0.unsqueeze(-1)).float()
        c0f = (be0[c].double() - mean0 * s0).float()
        W1 = P['W1'][c]
        pw, pb = _pack_single(W1, W0f, c0f, f)
        S = StatsFn.apply(x, W0f, c0f, W1, pw, pb, pat, f)
        if group_sum is not None:
            S = group_sum(S)
        Sd = S.double()
        m1 = Sd[..., 0] / n
        v1 = (Sd[..., 1] / n - m1 * m1).clamp_min(0.0)
        s1 = (1.0 / torch.sqrt(v1 + BN_EPS)).float()
        as1 = a[:, c] * s1
        cvec = -m1.float() + bsh[:, c] / as1
        u = P['W2'][c].unsqueeze(0) * as1.unsqueeze(2)
        x, lv, mu = ApplyFn.apply(x, W0f, c0f, W1, cvec, u, P['b2'][c], pw, pb, pat, f, eps, mode)
        ps[c], mus[c], lvs[c] = x, mu, lv
        logdet = lv if logdet is None else logdet + lv
        with torch.no_grad():
            unb = n / max(n - 1.0, 1.0)
            bn_batch[c, :, 0, 0] = mean0.float()
            bn_batch[c, :, 0, 1] = (var0 * unb).float()
            bn_batch[c, :, 1, 0] = m1.float()
            bn_batch[c, :, 1, 1] = (v1 * unb).float()
    return x, logdet, (ps, mus, lvs), bn_batch


# ======================================================================================================================
# The train-mode density pass of K stacks at once through the K-batched, phase-split C pipeline (csrc/gwtf_train.hip,
# include/gwtf.h GwtfTrainCtx): every kernel of the per-coupling chain above (folds and their backward included) takes all K
# mixture components in one launch, and the chain is cut exactly where a data-parallel run sums BatchNorm statistics over
# the ranks -- TWO collectives per depth level and direction for all K components and both branches (66 per forward of a
# 33-coupling config; reference: SyncBatchNorm, train_ae.py:152, 8 layers x 33 couplings x K collectives).  A single rank
# runs everything from two C calls.  Only the FiLM heads (O(B f G), vectorised over couplings) stay a torch graph.  The
# chain of autograd nodes above remains as the cross-check (tests: test_train_fast_path_equals_autograd_chain and the
# 2-rank test).
# ======================================================================================================================
COLLECTIVES = {'n': 0}        # statistic all-reduces issued by TrainMixtureFn (tests assert the count)
GRAD_SINK = {'reducer': None}  # set by dist.OverlappedGradients: receives the decoders' flat gradient as soon as it exists


def _stat_sum(t):
    """Sum a COMPACT statistic record over the ranks, in place.  In a data-parallel run every forward phase of the pipeline ends
    with one small launch that adds up the 64 copies its atomics were spread over into a contiguous [K][n] record (csrc/gwtf_train.hip
    stat_compact_kernel): that record goes on the wire (3 KB at f = 37, K = 4, where the copies are 196 KB) and every consumer --
    the folds, the backward pass -- reads it in place of the copies: no write-back, no second launch."""
    import torch.distributed as dist
    from .dist import run
    COLLECTIVES['n'] += 1
    run(dist.all_reduce, t, op=dist.ReduceOp.SUM)


def _zero_arena(shapes, device):
    """{name: zero tensor of shapes[name]} carved out of ONE zero-filled allocation (each region starts on a 256-byte boundary)."""
    offs, total = {}, 0
    for name, shape in shapes.items():
        n = 1
        for d in shape:
            n *= int(d)
        offs[name] = (total, n)
        total += (n + 63) // 64 * 64
    arena = torch.zeros(total, device=device, dtype=torch.float32)
    return {name: arena[o:o + n].view(*shapes[name]) for name, (o, n) in offs.items()}


class TrainMixtureFn(torch.autograd.Function):
    """out, logdet (K,B,3,N), lists (3,K,C,B,3,N), bn_batch (K,C,2,4,2,f) of K stacks with batch-statistic BatchNorm.
    raw (K, C*R): the stacks' raw arenas; film_raw (B, K*C, 2, 2, FP): raw FiLM {scale a, shift b} of this rank's shapes, zero
    beyond the f valid columns (FilmHeadsFn).  n_total: number of points the statistics cover over all ranks; sharded: statistics
    are all-reduced between phases."""

    @staticmethod
    def forward(ctx, p, raw, film_raw, K, C, f, G, pattern0, eps, mode, n_total, sharded, want_lists=True, handover=None):
        ctx.handover = handover            # see FilmHeadsFn.forward
        L = _lib.lib()
        p, raw = p.contiguous(), raw.contiguous()
        B, _, N = p.shape
        dev = p.device
        FP = L.gwtf_padded_width(f)
        FS, PB = L.gwtf_film_out_floats(f), L.gwtf_packed_b_coupling_floats(f)
        R = _lib.STAT_REPLICAS
        new = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
        film_raw = film_raw.contiguous()
        assert film_raw.shape == (B, K * C, 2, 2, FP), film_raw.shape
        # the zero-initialised buffers (accumulators of the passes' atomics, the padded W1^T image) from ONE fill
        shapes = dict(pb=(K * C * PB,), moments=(C + 1, K, R * 16), ystats=(C, K, R * 2 * FP * 2), bn_batch=(K, C, 2, 4, 2, f))
        zero = _zero_arena(shapes, dev)
        with torch.cuda.device(dev):
            pw, _ = _lib.pack_weights(raw.view(-1), C, f, G, True, pattern0, K=K, stack_only=True)
            pb = zero['pb']
            _lib.check(L.gwtf_pack_w1t(raw.data_ptr(), pb.data_ptr(), K * C, f, G, _lib._stream(p)))
        t = _lib.TrainCtx()
        t.K, t.B, t.N, t.C, t.f, t.G, t.pattern0, t.mode = K, B, N, C, f, G, pattern0, _lib._MODES[mode]
        t.eps, t.n_total, t.tune = float(eps), float(n_total), _lib.tune_word()
        bufs = dict(moments=zero['moments'], ystats=zero['ystats'], bn_batch=zero['bn_batch'],
                    # data parallel: the compact statistic records that are all-reduced and that every consumer then reads
                    mom_c=new(C + 1, K, 16) if sharded else None, ys_c=new(C, K, 2 * FP * 2) if sharded else None,
                    film_rec=new(B, K * C, FS), xbuf=new(2, K, B, 3, N), logdet=new(K, B, 3, N),
                    # the per-coupling inputs (ps) are what the backward recomputes from; mus / logvars only when the caller
                    # wants the reference's lists (36 B per point, coupling and component less to write otherwise)
                    lists=new(3 if want_lists else 1, K, C, B, 3, N))
        t.p, t.raw, t.packed_w, t.packed_b, t.film_raw = p.data_ptr(), raw.data_ptr(), pw.data_ptr(), pb.data_ptr(), film_raw.data_ptr()
        for name in ('moments', 'ystats', 'mom_c', 'ys_c', 'bn_batch', 'film_rec', 'xbuf', 'logdet'):
            setattr(t, name, bufs[name].data_ptr() if bufs[name] is not None else None)
        lists = bufs['lists']
        t.ps = lists[0].data_ptr()
        t.mus, t.logvars = (lists[1].data_ptr(), lists[2].data_ptr()) if want_lists else (None, None)
        t.stream = _lib._stream(p)
        with torch.cuda.device(dev):
            if not sharded:
                _lib.check(L.gwtf_mtrain_forward(ctypes.addressof(t)))
            else:
                mom, ys = bufs['mom_c'], bufs['ys_c']
                _lib.check(L.gwtf_mtrain_phase(ctypes.addressof(t), _lib.PHASE_FWD_INIT, 0))
                _stat_sum(mom[0, 0])                           # the shared input clouds' moments: one record
                for step in range(C):
                    c = step if mode == 'direct' else C - 1 - step
                    _lib.check(L.gwtf_mtrain_phase(ctypes.addressof(t), _lib.PHASE_FWD_A, step))
                    _stat_sum(ys[c])
                    _lib.check(L.gwtf_mtrain_phase(ctypes.addressof(t), _lib.PHASE_FWD_B, step))
                    if step + 1 < C:
                        _stat_sum(mom[step + 1])
        # the half of xbuf that holds the final coordinates, as a tensor of its own on the same storage (not a view: consumers that
        # look for the (K, ...) tensor behind K per-component slices -- models._restack -- find this one); the other half, 12 B per
        # point and component, stays allocated with it
        xb = bufs['xbuf']
        out = torch.empty(0, device=dev, dtype=torch.float32).set_(
            xb.untyped_storage(), xb.storage_offset() + L.gwtf_mtrain_final_forward_half(C) * K * B * 3 * N, (K, B, 3, N))
        # the backward reads the forward statistics again: the compact (all-reduced) records when data parallel, the copies otherwise
        ctx.save_for_backward(p, raw, lists[0], pw, pb, bufs['film_rec'], film_raw,
                              bufs['mom_c' if sharded else 'moments'], bufs['ys_c' if sharded else 'ystats'])
        ctx.meta = (int(K), int(C), int(f), int(G), int(FP), int(pattern0), float(eps), mode, float(n_total), bool(sharded))
        ctx.mark_non_differentiable(bufs['bn_batch'])
        ctx.set_materialize_grads(False)
        # ps / logvars list slots are differentiable (the backward kernels add a slot's gradient where its coupling is
        # processed); a gradient through a mus slot raises
        if not want_lists:
            empty = lists.new_empty(0)
            return out, bufs['logdet'], lists[0], empty, empty, bufs['bn_batch']
        return out, bufs['logdet'], lists[0], lists[1], lists[2], bufs['bn_batch']

    @staticmethod
    def backward(ctx, g_out, g_logdet, g_ps, g_mus, g_lvs, _gb):
        if g_mus is not None:
            raise NotImplementedError('a gradient reached a mus[j] list entry: the HIP backward differentiates through ps[j], '
                                      'logvars[j], the final coordinates and sum(logvars) only')
        p, raw, ps_saved, pw, pb, film_rec, film_raw, mom, ystats = ctx.saved_tensors
        K, C, f, G, FP, pattern0, eps, mode, n_total, sharded = ctx.meta
        L = _lib.lib()
        B, _, N = p.shape
        dev = p.device
        R = _lib.STAT_REPLICAS
        new = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
        zeros = lambda *shape: torch.zeros(*shape, device=dev, dtype=torch.float32)
        g_out = (g_out if g_out is not None else zeros(K, B, 3, N)).contiguous().float()
        g_ld = (g_logdet if g_logdet is not None else zeros(K, B, 3, N)).contiguous().float()
        t = _lib.TrainCtx()
        t.K, t.B, t.N, t.C, t.f, t.G, t.pattern0, t.mode = K, B, N, C, f, G, pattern0, _lib._MODES[mode]
        t.eps, t.n_total, t.tune = eps, n_total, _lib.tune_word()
        scratch = new(K, B, 3, N)
        # (one fill for the accumulators; the two gradients that leave this function keep their own storage)
        zero = _zero_arena(dict(g_film=(B, K * C, 2, 3, FP), g_sd0=(C, K, R * 2 * 3 * FP), g_bias=(C, K, R * 4), g_mom=(C, K, 16)), dev)
        bufs = dict(g_bufs=new(2, K, B, 3, N),
                    dw1_ws=new(K * L.gwtf_mtrain_dw1_floats(f, B, N)), g_film=zero['g_film'],
                    g_sd0=zero['g_sd0'], g_bias=zero['g_bias'], g_stats=new(C, K, 2 * 2 * FP),
                    g_mom=zero['g_mom'], g_film_raw=torch.zeros_like(film_raw),
                    g_raw=torch.zeros_like(raw))
        t.p, t.raw, t.packed_w, t.packed_b = p.data_ptr(), raw.data_ptr(), pw.data_ptr(), pb.data_ptr()
        t.film_raw, t.film_rec = film_raw.data_ptr(), film_rec.data_ptr()
        if sharded:
            t.mom_c, t.ys_c = mom.data_ptr(), ystats.data_ptr()
        else:
            t.moments, t.ystats = mom.data_ptr(), ystats.data_ptr()
        t.bn_batch, t.xbuf, t.logdet = bufs['g_stats'].data_ptr(), bufs['g_bufs'].data_ptr(), scratch.data_ptr()   # forward-only fields
        t.ps, t.mus, t.logvars = ps_saved.data_ptr(), ps_saved.data_ptr(), ps_saved.data_ptr()
        t.g_out, t.g_ld = g_out.data_ptr(), g_ld.data_ptr()
        g_ps = g_ps.contiguous().float() if g_ps is not None else None
        g_lvs = g_lvs.contiguous().float() if g_lvs is not None else None
        t.g_ps = g_ps.data_ptr() if g_ps is not None else None
        t.g_lvs = g_lvs.data_ptr() if g_lvs is not None else None
        for name, buf in bufs.items():
            setattr(t, name, buf.data_ptr())
        t.stream = _lib._stream(p)
        with torch.cuda.device(dev):
            if not sharded:
                _lib.check(L.gwtf_mtrain_backward(ctypes.addressof(t)))
            else:
                for step in range(C):
                    c = step if mode == 'inverse' else C - 1 - step
                    _lib.check(L.gwtf_mtrain_phase(ctypes.addressof(t), _lib.PHASE_BWD_A, step))
                    _stat_sum(bufs['g_stats'][c])
                    _lib.check(L.gwtf_mtrain_phase(ctypes.addressof(t), _lib.PHASE_BWD_B, step))
                    _stat_sum(bufs['g_mom'][c])
                    _lib.check(L.gwtf_mtrain_phase(ctypes.addressof(t), _lib.PHASE_BWD_C, step))
        dp = bufs['g_bufs'][L.gwtf_mtrain_final_backward_half(C, _lib._MODES[mode])]
        dp = dp[0] if K == 1 else dp.sum(0)                    # the K components read the same clouds
        g_raw = bufs["g_raw"]
        if ctx.handover is not None and ctx.needs_input_grad[2] and ctx.needs_input_grad[1]:
            ctx.handover['g_raw'] = g_raw          # the FiLM heads' backward adds its slots and returns the one buffer
            g_raw = None
        return (dp, g_raw, bufs['g_film_raw']) + (None,) * 11


def train_density_forward_multi(engines, p, g, mode='inverse', distributed=False, want_lists=True):
    """Train-mode density pass of K stacks (the components of a mixture, or one decoder) through the fused pipeline.
    -> out, logdet (K,B,3,N), lists = (ps, mus, lvs) each (K,C,B,3,N) with ps / lvs differentiable, bn_batch (K,C,2,4,2,f)
    incl. the FiLM BatchNorm statistics."""
    e0 = engines[0]
    K, C, f, G, eps = len(engines), e0.C, e0.f, e0.G, e0.couplings[0]._eps_value
    B, _, N = p.shape
    g = g.float()
    row0, rows_total = 0, B
    if distributed:
        # the latents of all ranks (differentiable all-gather; uneven per-rank batches allowed, train_ae.py:77-78): the per-shape
        # FiLM BatchNorm normalises with the global batch.  The per-rank row counts are cached (dist.row_layout): no host sync.
        from .dist import gather_rows
        g_all, lay = gather_rows(g)
        row0, rows_total = lay.row0, lay.total
    else:
        g_all = g
    if rows_total < 2:
        raise ValueError('train-mode BatchNorm needs more than 1 shape per (global) batch (torch raises the same)')
    from .flows import stacked_raw_arena
    raw = stacked_raw_arena(engines)                          # (K, R): one gather launch, one autograd node
    # the FiLM heads of all K stacks (4 K C small MLPs, BatchNorm over the latent rows of ALL ranks): one HIP launch forward, two
    # backward, parameters read from / gradients written into the stacked arena in place (csrc/gwtf_film_train.hip)
    # (any number of latent rows: the kernels walk them 128 at a time)
    handover = {} if raw.requires_grad else None
    film_raw, film_mean, film_var = FilmHeadsFn.apply(raw, g_all, K * C, f, G, row0, B, eps, True, handover)
    film_var = film_var * (rows_total / max(rows_total - 1.0, 1.0))  # unbiased, as BatchNorm's running_var update takes it
    if raw.requires_grad:
        # every stack's whole parameter gradient is ONE flat tensor (the gradient of its raw arena: the pipeline's part plus
        # the FiLM heads' part, summed by autograd): hand it to the data-parallel reducer (dist.OverlappedGradients, looked up
        # when the backward pass runs) the moment it exists, so that its all-reduce overlaps the rest of the backward pass
        def _to_reducer(grad, engines=tuple(engines)):
            sink = GRAD_SINK['reducer']
            if sink is not None:
                for k, e in enumerate(engines):
                    sink.on_flat_gradient(grad[k], e)
        raw.register_hook(_to_reducer)
    out, logdet, ps, mus, lvs, bn_batch = TrainMixtureFn.apply(p.float(), raw, film_raw, K, C, f, G, e0.pattern0, eps, mode,
                                                               float(rows_total) * N, distributed, want_lists, handover)
    lists = (ps, mus, lvs) if want_lists else None           # (K,C,B,3,N) each; ps / lvs differentiable
    bn_batch = bn_batch.clone()
    bn_batch[:, :, :, 2:4, 0] = film_mean.view(K, C, 2, 2, f)
    bn_batch[:, :, :, 2:4, 1] = film_var.view(K, C, 2, 2, f)
    return out, logdet, lists, bn_batch


def train_density_forward_fast(engine, p, g, mode='inverse', distributed=False):
    """One stack through the fused pipeline.  -> out, logdet (B,3,N), lists = (ps, mus, lvs) (C,B,3,N) each, bn_batch (C,2,4,2,f)."""
    out, logdet, lists, bn_batch = train_density_forward_multi([engine], p, g, mode, distributed)
    return out[0], logdet[0], tuple(t[0] for t in lists), bn_batch[0]
