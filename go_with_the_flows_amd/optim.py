"""Fused Adam / AMSGrad with the reference's semantics (lib/networks/optimizers.py:8-76).

Same constructor, same ``state`` entries ('step', 'exp_avg', 'exp_avg_sq', 'max_exp_avg_sq') -- so
``optimizer_state`` checkpoints written by the reference (training.py:76-81) load with ``load_state_dict`` -- and the
same update, including its un-scaled decoupled weight decay ``p -= wd*p + lr*m_hat/(sqrt(v_hat)+eps)``.  The
per-parameter Python loop (about a dozen tiny launches per tensor) becomes one HIP launch per 48 tensors
(csrc/gwtf_adam.hip).  ``LRUpdater`` (optimizers.py:79-97) works unchanged: it only edits ``param_groups``.
"""
import ctypes

import torch
from torch.optim import Optimizer

from . import _lib


class Adam(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad))
        self._tables = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        for group in self.param_groups:
            ams = bool(group['amsgrad'])
            by_step = {}
            for p in group['params']:
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
                    if ams:
                        st['max_exp_avg_sq'] = torch.zeros_like(p)
                st['step'] += 1
                by_step.setdefault((int(st['step']), p.device), []).append(p)
            for (step, dev), plist in by_step.items():
                n = len(plist)
                arr = ctypes.c_void_p * n
                # parameter / state pointers are stable across steps: cache the tables, refresh only the gradients
                key = (id(group), step > 1, dev, n, plist[0].data_ptr(), plist[-1].data_ptr(), ams)
                tabs = self._tables.get(key)
                if tabs is None:
                    sts = [self.state[p] for p in plist]
                    ptr = lambda ts: arr(*[t.data_ptr() for t in ts])
                    tabs = (ptr(plist), ptr([s['exp_avg'] for s in sts]), ptr([s['exp_avg_sq'] for s in sts]),
                            ptr([s['max_exp_avg_sq'] for s in sts]) if ams else None,
                            (ctypes.c_size_t * n)(*[p.numel() for p in plist]))
                    self._tables = {key: tabs} if len(self._tables) > 8 else {**self._tables, key: tabs}
                grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in plist]
                gptr = arr(*[t.data_ptr() for t in grads])
                b1, b2 = group['betas']
                with torch.cuda.device(dev):
                    _lib.check(L.gwtf_adam_step(tabs[0], gptr, tabs[1], tabs[2], tabs[3], tabs[4], n, float(group['lr']),
                                                float(b1), float(b2), float(group['eps']), float(group['weight_decay']), step,
                                                int(ams), torch.cuda.current_stream(dev).cuda_stream))
        return loss
