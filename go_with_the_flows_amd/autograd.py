"""Differentiable density pass (inverse direction) of a coupling stack: HIP forward + HIP backward.

Structure: the O(f^2 + B*f*G) "fold" that turns module parameters and the latent g into
the folded quantities the kernels consume (BatchNorm folded into sd0/sd1, FiLM heads -> per-shape {c, u}) is a
small torch graph -- elementwise ops and library GEMMs on (C,2,f,*) / (B,C,2,f) tensors, the same arithmetic as
csrc/gwtf_pack.hip + csrc/gwtf_film.hip -- so autograd carries gradients from the folded quantities back to every
reference parameter and to g.  Everything per point (the forward stack, and per coupling the recompute + dacc +
dh = W1p^T dacc + sd0/FiLM-record gradient reductions + dx) is hand-written HIP (csrc/gwtf_stack.hip,
csrc/gwtf_bwd.hip).  The f x f weight gradient sum_p dacc(p) h(p)^T is a plain GEMM over all points and goes
to the BLAS library (torch.einsum -> rocBLAS).

Scope: BatchNorm as a fixed affine (model.eval(): running statistics).  Batch-statistic BatchNorm adds
gradient terms through the statistics; that backward is not built yet and train mode raises at backward().
Reference semantics: loss.backward() through LocalCondRNVPDecoder.forward(mode='inverse'), training.py:54.
"""
import torch
import torch.nn.functional as F

from . import _lib

BN_EPS = 1e-5


def _gather(engine):
    """Stack the couplings' parameters / buffers into (C,2,...) tensors, branch order (logvar, mu)."""
    cps = engine.couplings
    f = engine.f

    def st(fn):
        return torch.stack([torch.stack([fn(c, X) for X in ('logvar', 'mu')]) for c in cps])

    def pad_last(t, n):
        return t if t.shape[-1] == n else F.pad(t, (0, n - t.shape[-1]))

    def pad_rows(t, n):
        return t if t.shape[0] == n else F.pad(t, (0, 0, 0, n - t.shape[0]))

    T0 = lambda c, X: getattr(c, f'T_{X}_0')
    head = lambda c, X, w: getattr(c, f'T_{X}_0_cond_{w}')
    out = {
        'W0': st(lambda c, X: pad_last(T0(c, X)[0].weight[0], 2)),            # (C,2,f,2)
        'bn0': [st(lambda c, X, k=k: getattr(T0(c, X)[1], k)) for k in ('weight', 'bias', 'running_mean', 'running_var')],
        'W1': st(lambda c, X: T0(c, X)[3].weight[0]),                          # (C,2,f,f)
        'bn1': [st(lambda c, X, k=k: getattr(T0(c, X)[4], k)) for k in ('running_mean', 'running_var')],
        'W2': st(lambda c, X: pad_rows(getattr(c, f'T_{X}_1')[1].weight[0], 2)),   # (C,2,2,f)
        'b2': st(lambda c, X: pad_last(getattr(c, f'T_{X}_1')[1].bias[0], 2)),     # (C,2,2)
    }

    def st_heads(fn):
        return torch.stack([torch.stack([torch.stack([fn(head(c, X, w)) for w in ('w', 'b')]) for X in ('logvar', 'mu')])
                            for c in cps])

    out['L0'] = st_heads(lambda h: h[0].weight)                                # (C,2,2,f,G)
    out['hbn'] = [st_heads(lambda h, k=k: getattr(h[1], k)) for k in ('weight', 'bias', 'running_mean', 'running_var')]
    out['L1'] = st_heads(lambda h: h[3].weight)                                # (C,2,2,f,f)
    out['b1'] = st_heads(lambda h: h[3].bias)                                  # (C,2,2,f)
    return out


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
    hg, hb, hrm, hrv = P['hbn']
    S = hg / torch.sqrt(hrv + BN_EPS)
    T = hb - hrm * S
    hraw = torch.einsum('bg,cxhfg->bcxhf', g, P['L0'])
    hbn = hraw * S + T
    hn = hbn * torch.sigmoid(hbn)
    o = torch.einsum('bcxhi,cxhji->bcxhj', hn, P['L1']) + P['b1']
    a = eps + torch.exp(o[:, :, :, 0])
    cvec = c1 + o[:, :, :, 1] / a
    u = P['W2'].unsqueeze(0) * a.unsqueeze(3)
    return W0f, c0f, W1p, cvec, u, P['b2']


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
    """out, logdet = inverse pass of the whole stack, differentiable w.r.t. p and the folded tensors."""

    @staticmethod
    def forward(ctx, p, W0f, c0f, W1p, cvec, u, b2, C, f, pattern0, eps):
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
        out, logdet, lists = _lib.stack_forward(p, pw, rec, C, f, pattern0, eps, 'inverse', True)
        ctx.save_for_backward(p, lists[0], pw, pb, rec)
        ctx.meta = (C, f, FP, pattern0, eps)
        return out, logdet

    @staticmethod
    def backward(ctx, g_out, g_logdet):
        p, ps, pw, pb, rec = ctx.saved_tensors
        C, f, FP, pattern0, eps = ctx.meta
        L = _lib.lib()
        B, _, N = p.shape
        dev = p.device
        R = _lib.STAT_REPLICAS
        PW, PB = L.gwtf_packed_w_coupling_floats(f), L.gwtf_packed_b_coupling_floats(f)
        cur = (g_out if g_out is not None else torch.zeros_like(p)).contiguous().float()
        g_ld = (g_logdet if g_logdet is not None else torch.zeros_like(p)).contiguous().float()
        g_film = torch.zeros(B, C, 2, 3, FP, device=dev, dtype=torch.float32)
        g_sd0 = torch.zeros(C, R, 2, 3, FP, device=dev, dtype=torch.float32)
        g_bias = torch.zeros(C, R, 4, device=dev, dtype=torch.float32)
        dA = torch.empty(B, 2, FP, N, device=dev, dtype=torch.float32)
        H0 = torch.empty(B, 2, FP, N, device=dev, dtype=torch.float32)
        gW1p = torch.empty(C, 2, f, f, device=dev, dtype=torch.float32)
        bufs = [torch.empty_like(p), torch.empty_like(p)]
        st = _lib._stream(p)
        with torch.cuda.device(dev):
            for c in range(C):                       # reverse of the forward's processing order C-1 .. 0
                x_in = ps[c + 1] if c + 1 < C else p
                nxt = bufs[c & 1]
                _lib.check(L.gwtf_coupling_backward(x_in.data_ptr(), cur.data_ptr(), g_ld.data_ptr(),
                                                    pw[c * PW:].data_ptr(), pb[c * PB:].data_ptr(), rec.data_ptr(),
                                                    nxt.data_ptr(), dA.data_ptr(), H0.data_ptr(), g_film.data_ptr(),
                                                    g_sd0[c].data_ptr(), g_bias[c].data_ptr(), c, B, N, C, f, pattern0,
                                                    float(eps), st))
                # dW1p[k][j][i] = sum_{b,n} dacc[b,k,j,n] h[b,k,i,n]: plain GEMM -> BLAS
                gW1p[c] = torch.einsum('bkjn,bkin->kji', dA[:, :, :f], H0[:, :, :f])
                cur = nxt
        gs = g_sd0.sum(1)                                              # (C,2,3,FP)
        g_W0f = gs[:, :, 0:2, :f].permute(0, 1, 3, 2).contiguous()    # (C,2,f,2)
        g_c0f = gs[:, :, 2, :f].contiguous()
        g_cvec = g_film[:, :, :, 0, :f].contiguous()
        g_u = g_film[:, :, :, 1:3, :f].contiguous()
        g_b2 = g_bias.sum(1).reshape(C, 2, 2)
        return cur, g_W0f, g_c0f, gW1p, g_cvec, g_u, g_b2, None, None, None, None


def density_forward(engine, p, g):
    """Differentiable (out, logdet) of the inverse pass; eval-mode BatchNorm."""
    eps = engine.couplings[0]._eps_value
    W0f, c0f, W1p, cvec, u, b2 = fold(engine, g.float(), eps)
    return StackDensityFn.apply(p.float(), W0f, c0f, W1p, cvec, u, b2, engine.C, engine.f, engine.pattern0, eps)
