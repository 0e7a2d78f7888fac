"""Fused Adam / AMSGrad with the reference's semantics (lib/networks/optimizers.py:8-76).

Same constructor, same ``state`` entries ('step', 'exp_avg', 'exp_avg_sq', 'max_exp_avg_sq') -- so
``optimizer_state`` checkpoints written by the reference (training.py:76-81) load with ``load_state_dict`` -- and the
same update, including its un-scaled decoupled weight decay ``p -= wd*p + lr*m_hat/(sqrt(v_hat)+eps)``.  The
per-parameter Python loop (about a dozen tiny launches per tensor) becomes ONE HIP launch for all tensors of a parameter
group (csrc/gwtf_adam.hip: pointer table in device memory; up to 48 tensors ride in the kernel arguments instead).  ``LRUpdater`` (optimizers.py:79-97) works unchanged: it only edits ``param_groups``.
"""
import ctypes

import torch
from torch.optim import Optimizer

from . import _lib


class Adam(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad))
        self._plans = {}

    def load_state_dict(self, state_dict):
        self._plans = {}
        return super().load_state_dict(state_dict)

    def _make_plan(self, plist, dev, step, ams):
        """Pointer tables of parameters and moments (stable across steps) for one launch group."""
        n = len(plist)
        arr = ctypes.c_void_p * n
        sts = [self.state[p] for p in plist]
        ptr = lambda ts: arr(*[t.data_ptr() for t in ts])
        plan = {'n': n, 'arr': arr, 'dev': dev, 'step': step, 'ams': ams, 'states': sts, 'p0': plist[0].data_ptr(), 'plist': plist,
                'ptrs': tuple(t.data_ptr() for t in plist),
                'p': ptr(plist), 'm': ptr([s['exp_avg'] for s in sts]), 'v': ptr([s['exp_avg_sq'] for s in sts]),
                'vmax': ptr([s['max_exp_avg_sq'] for s in sts]) if ams else None,
                'numel': (ctypes.c_size_t * n)(*[p.numel() for p in plist])}
        if n > self._TABLE_ABOVE:
            # many tensors: ONE launch with the pointer table in device memory (built once; only the gradient pointers are
            # re-uploaded, and only when they moved)
            chunk = _lib.lib().gwtf_adam_chunk_elems()
            rows = [[p.data_ptr(), s['exp_avg'].data_ptr(), s['exp_avg_sq'].data_ptr(),
                     s['max_exp_avg_sq'].data_ptr() if ams else 0, p.numel()] for p, s in zip(plist, sts)]
            cmap = [(i, c) for i, p in enumerate(plist) for c in range((p.numel() + chunk - 1) // chunk)]
            plan['table'] = torch.tensor(rows, dtype=torch.int64).to(dev)
            plan['cmap'] = torch.tensor(cmap, dtype=torch.int32).reshape(-1, 2).to(dev)
            plan['n_chunks'] = len(cmap)
            plan['gtab'] = torch.zeros(n, dtype=torch.int64, device=dev)
            plan['gptrs'] = None
            # pinned staging (two buffers, an event each): the upload must not synchronise the host with the stream
            plan['gstage'] = [torch.zeros(n, dtype=torch.int64).pin_memory() for _ in range(2)]
            plan['gevent'] = [None, None]
            plan['gflip'] = 0
        return plan

    _TABLE_ABOVE = 48   # tensors per launch of the pointer-argument kernel (csrc/gwtf_adam.hip kMaxT)

    @staticmethod
    def _bump_versions(plist):
        """The kernels write parameters through raw pointers: tell autograd / the packed-weight caches (which key on tensor
        version counters, flows.StackEngine._key, encoders.packed) that the values changed.  No kernel is launched."""
        bump = torch._C._increment_version
        try:
            bump(plist)                    # torch >= 2.4 takes an iterable
        except TypeError:
            for p in plist:
                bump(p)

    @staticmethod
    def _launch(L, group, plan, grads, step):
        b1, b2 = group['betas']
        dev = plan['dev']
        grads = [g if g.is_contiguous() else g.contiguous() for g in grads]   # copies stay alive until the launch is enqueued
        if 'table' in plan:
            gp = [g.data_ptr() for g in grads]
            if gp != plan['gptrs']:
                k = plan['gflip'] = plan['gflip'] ^ 1
                if plan['gevent'][k] is not None:
                    plan['gevent'][k].synchronize()          # the copy that last read this staging buffer (two steps ago)
                plan['gstage'][k].numpy()[:] = gp
                with torch.cuda.device(dev):
                    plan['gtab'].copy_(plan['gstage'][k], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(dev))
                plan['gevent'][k] = ev
                plan['gptrs'] = gp
            with torch.cuda.device(dev):
                _lib.check(L.gwtf_adam_step_table(plan['table'].data_ptr(), plan['gtab'].data_ptr(), plan['cmap'].data_ptr(),
                                                  plan['n_chunks'], float(group['lr']), float(b1), float(b2),
                                                  float(group['eps']), float(group['weight_decay']), step, int(plan['ams']),
                                                  torch.cuda.current_stream(dev).cuda_stream))
            return
        gptr = plan['arr'](*[g.data_ptr() for g in grads])
        with torch.cuda.device(dev):
            _lib.check(L.gwtf_adam_step(plan['p'], gptr, plan['m'], plan['v'], plan['vmax'], plan['numel'], plan['n'],
                                        float(group['lr']), float(b1), float(b2), float(group['eps']),
                                        float(group['weight_decay']), step, int(plan['ams']),
                                        torch.cuda.current_stream(dev).cuda_stream))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        for gi, group in enumerate(self.param_groups):
            ams = bool(group['amsgrad'])
            params = group['params']
            plan = self._plans.get(gi)
            # fast path: the same parameters as last step carry gradients (and only those), one common step count -> reuse
            # the pointer tables; only the step counters and the gradient pointers are touched
            if plan is not None and plan['ams'] == ams and plan['n_group'] == len(params) and \
                    plan['ptrs'] == tuple(params[i].data_ptr() for i in plan['idx']):
                all_grads = [p.grad for p in params]
                grads = [all_grads[i] for i in plan['idx']]
                if all(g is not None for g in grads) and sum(g is not None for g in all_grads) == plan['n']:
                    step = plan['step'] = plan['step'] + 1
                    for st in plan['states']:
                        st['step'] = step
                    self._launch(L, group, plan, grads, step)
                    self._bump_versions(plan['plist'])
                    continue
            self._plans.pop(gi, None)
            by_step = {}
            for i, p in enumerate(params):
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError('Adam does not support sparse gradients')
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise _lib.GwtfError('fused Adam needs contiguous float32 parameters on a HIP device (no CPU path)')
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p)
                    st['exp_avg_sq'] = torch.zeros_like(p)
                if ams and 'max_exp_avg_sq' not in st:
                    st['max_exp_avg_sq'] = torch.zeros_like(p)
                st['step'] += 1
                by_step.setdefault((int(st['step']), p.device), []).append(i)
            for (step, dev), idx in by_step.items():
                plist = [params[i] for i in idx]
                plan = self._make_plan(plist, dev, step, ams)
                plan['idx'], plan['n_group'], plan['p0'] = idx, len(params), params[0].data_ptr()
                self._launch(L, group, plan, [p.grad for p in plist], step)
                self._bump_versions(plist)
                if len(by_step) == 1:
                    self._plans[gi] = plan
        return loss


class LRUpdater:
    """Cosine cycle of the learning rate and of beta2 (reference optimizers.py:79-97): within a cycle of ``cycle_length``
    epochs both go from their max to their min; beta1 is fixed.  Called as ``updater(optimizer, epoch, iteration)``."""

    def __init__(self, epoch_length, **kwargs):
        self.epoch_length = epoch_length
        self.cycle_length = kwargs['cycle_length']
        self.min_lr, self.max_lr = kwargs['min_lr'], kwargs['max_lr']
        self.beta1 = kwargs['beta1']
        self.min_beta2, self.max_beta2 = kwargs['min_beta2'], kwargs['max_beta2']

    def __call__(self, optimizer, epoch, iteration):
        import math
        phase = ((epoch % self.cycle_length) * self.epoch_length + iteration) / (self.cycle_length * self.epoch_length)
        w = 0.5 * (1.0 + math.cos(math.pi * phase))
        for group in optimizer.param_groups:
            group['lr'] = self.min_lr + (self.max_lr - self.min_lr) * w
            group['betas'] = (self.beta1, self.min_beta2 + (self.max_beta2 - self.min_beta2) * w)
