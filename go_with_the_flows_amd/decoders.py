"""Point-cloud decoder: a stack of ``n_flows`` coupling Triples run as ONE fused HIP launch.

Drop-in for the reference's ``LocalCondRNVPDecoder`` (lib/networks/decoders.py:41-79): same
constructor, attributes, ``state_dict`` keys, static ``get_param_count`` and
``forward(p, g, mode) -> (ps, mus, logvars)`` (direct-ordered lists of 3*n_flows ``(B,3,N)``
tensors).  ``forward_fused`` returns only what the reference's consumers read -- final
coordinates and the per-coordinate sum of logvars (losses.py:112-116, flow_mixture.py:175) --
and skips materialising the 9*n_flows intermediate tensors.

Sibling batching (the literal import swap of INTEGRATION.md section 1).  The reference keeps its K mixture components in an
``nn.ModuleList`` and calls them ONE AT A TIME on the same ``(p, g)`` (flow_mixture.py:163-166).  In train mode every such call
is a chain of 8 launches per depth level whose small kernels do not fill the GPU at K = 1; the K-batched pipeline
(csrc/gwtf_train.hip) runs the same chain once for all K.  A decoder therefore remembers the ``ModuleList`` it was registered in
(a module-registration hook: the parent is not otherwise known to a child), and once a full round of sibling calls on one
``(p, g)`` has been OBSERVED, the first call of the next round runs all K components in one pass and the later calls return
their slices.  Nothing observable changes: each decoder's BatchNorm running statistics are updated when ITS call arrives, the
lists are differentiable exactly as the per-decoder ones, and a round that is not completed (a caller that stops after one
component) costs the speculated work once and switches the group back to per-decoder calls.
"""
import os
import weakref

import torch
import torch.nn as nn
from torch.nn.modules.module import register_module_module_registration_hook

from .flows import CondRealNVPFlow3DTriple, StackEngine


class _SiblingGroup:
    """K same-shaped decoders of one ``nn.ModuleList``: observes the call pattern, batches a round once it has seen one."""

    def __init__(self, decoders):
        self.decoders = list(decoders)
        self.index = {id(d): k for k, d in enumerate(self.decoders)}
        self.K = len(self.decoders)
        self.stack = None
        self.confirmed = False          # a complete round of sibling calls on one (p, g) has been seen
        self.seen_key, self.seen = None, set()
        self.round = None               # the speculated round: dict(key, p, g, out, logdet, lists, bn_batch, pending)
        self.stats = {'batched_rounds': 0, 'abandoned_rounds': 0, 'single_calls': 0}

    @staticmethod
    def key_of(p, g, mode):
        return (id(p), p._version, p.data_ptr(), tuple(p.shape), id(g), g._version, g.data_ptr(), mode, torch.is_grad_enabled())

    def applicable(self, p, g):
        d0 = self.decoders[0]
        if not (p.is_cuda and g.is_cuda) or p.dim() != 3 or p.shape[2] == 0 or os.environ.get('GWTF_NO_SIBLING_BATCH') == '1':
            return False
        if any(not d.training or (d.n_flows, d.f_n_features, d.g_n_features) != (d0.n_flows, d0.f_n_features, d0.g_n_features)
               for d in self.decoders):
            return False
        return not any(getattr(d.engine(), 'force_autograd_chain', False) for d in self.decoders)

    def lists_for(self, dec, p, g, mode):
        """The (ps, mus, logvars) lists of ``dec`` from a batched round, or None: the caller then runs its own pipeline."""
        k = self.index[id(dec)]
        if not self.applicable(p, g):
            self.round = None
            return None
        key = self.key_of(p, g, mode)
        rnd = self.round
        if rnd is not None and rnd['key'] == key and k in rnd['pending']:
            return self._take(rnd, k)
        if rnd is not None and rnd['pending']:
            # a new round began before every sibling had asked for its slice: the speculation was wrong for this caller
            self.stats['abandoned_rounds'] += 1
            self.confirmed, self.seen_key, self.seen = False, None, set()
        self.round = None
        if not self.confirmed:
            if key != self.seen_key or k in self.seen:
                self.seen_key, self.seen, self._seen_refs = key, set(), (p, g)      # (the references keep id() unique)
            self.seen.add(k)
            if len(self.seen) == self.K:
                self.confirmed, self.seen_key, self.seen, self._seen_refs = True, None, set(), None
            self.stats['single_calls'] += 1
            return None
        if self.stack is None:
            from .mixture import MixtureStack
            self.stack = MixtureStack(self.decoders)
        res = self.stack.forward_all_lists(p, g, mode, defer_running_stats=True)
        if res is None:
            return None
        out, logdet, lists, bn_batch = res
        self.stats['batched_rounds'] += 1
        # (one unbind per round: its backward stacks the K gradients with one launch; a select per call would each materialise a
        # zero-filled (K, B, 3, N) gradient and add them up)
        self.round = rnd = dict(key=key, p=p, g=g, out=out.unbind(0), logdet=logdet.unbind(0), lists=lists, bn_batch=bn_batch,
                                final=0 if mode == 'inverse' else lists[0].shape[1] - 1, pending=set(range(self.K)))
        return self._take(rnd, k)

    def _take(self, rnd, k):
        rnd['pending'].discard(k)
        e = self.decoders[k].engine()
        e._update_running_stats(rnd['bn_batch'][k])        # this decoder's BatchNorm buffers move when ITS call arrives
        e._last_lists = None
        ps, mus, lvs = rnd['lists']
        res = list(ps[k].unbind(0)), list(mus[k].unbind(0)), list(lvs[k].unbind(0))
        # the slot of the fully transformed cloud IS the pipeline's output tensor (same kernel, same registers): handing that tensor
        # out keeps the loss's gradient off the per-slot route, whose backward materialises a (K, C, B, 3, N) gradient per component
        res[0][rnd['final']] = rnd['out'][k]
        # the pipeline's own log-det of this component rides along on the logvars slots: a consumer that is about to add the C slots
        # up again (models.FlowMixtureNLL on the reference's lists) recognises the complete list and takes the sum that exists
        tag = _SlotSum(rnd['logdet'][k], len(res[2]))
        for j, t in enumerate(res[2]):
            t._gwtf_slot = (tag, j, t._version)
        if not rnd['pending']:
            self.round = None                               # nothing left to hand out: release the references
        return res


class _SlotSum:
    """sum of the C logvars slots of one decoder call, as the pipeline computed it (same values, one tensor instead of C - 1 adds and
    a gradient into every slot)"""
    __slots__ = ('total', 'C')

    def __init__(self, total, C):
        self.total, self.C = total, C


def slot_sum(tensors):
    """The stack's own sum of `tensors` when they are exactly the C logvars slots of one batched decoder call, in order; else None."""
    tag = getattr(tensors[0], '_gwtf_slot', (None, 0))[0] if tensors else None
    if tag is None or len(tensors) != tag.C:
        return None
    for j, t in enumerate(tensors):
        s = getattr(t, '_gwtf_slot', None)
        if s is None or s[0] is not tag or s[1] != j or s[2] != t._version:       # (an in-place edit of a slot voids the shortcut)
            return None
    return tag.total


def _note_parent(parent, name, child):
    """Module-registration hook: a decoder placed in an ``nn.ModuleList`` remembers the list (its siblings are found there)."""
    if isinstance(child, LocalCondRNVPDecoder) and isinstance(parent, nn.ModuleList):
        child.__dict__['_sibling_list'] = weakref.ref(parent)


class LocalCondRNVPDecoder(nn.Module):
    def __init__(self, n_flows, f_n_features, g_n_features, weight_std=0.01):
        super().__init__()
        self.n_flows, self.f_n_features, self.g_n_features = n_flows, f_n_features, g_n_features
        self.weight_std = weight_std
        self.flows = nn.ModuleList(
            CondRealNVPFlow3DTriple(f_n_features, g_n_features, weight_std=weight_std, pattern=i % 2)
            for i in range(n_flows))
        self._engine = None

    def __getstate__(self):
        """copy.deepcopy / pickle: the copy is not a member of the original's ModuleList (a weak reference is copied as itself)."""
        state = self.__dict__.copy()
        state.pop('_sibling_list', None)
        return state

    @staticmethod
    def get_param_count(n_flows, f_n_features, g_n_features):
        """The reference's sizing formula (decoders.py:54-59); it under-counts the true module on purpose."""
        per_coupling = 18 * f_n_features + 4 * f_n_features * g_n_features + 6 * f_n_features ** 2
        return n_flows * 3 * per_coupling

    def engine(self):
        if self._engine is None:
            self._engine = StackEngine([c for t in self.flows for c in t.couplings()])
        return self._engine

    def sibling_group(self):
        """The _SiblingGroup of the ``nn.ModuleList`` this decoder sits in (None: no list, a list of one, or other members)."""
        ref = self.__dict__.get('_sibling_list')
        parent = ref() if ref is not None else None
        if parent is None:
            return None
        members = list(parent)
        grp = parent.__dict__.get('_gwtf_sibling_group')
        if grp is None or len(grp.decoders) != len(members) or any(a is not b for a, b in zip(grp.decoders, members)):
            if len(members) < 2 or not all(isinstance(m, LocalCondRNVPDecoder) for m in members) or \
                    not any(m is self for m in members):
                parent.__dict__.pop('_gwtf_sibling_group', None)
                return None
            grp = parent.__dict__['_gwtf_sibling_group'] = _SiblingGroup(members)
        return grp

    def forward(self, p, g, mode='direct'):
        """Reference contract (decoders.py:61-79).  ``ps[0]`` is the base-space point after a full
        inverse, ``ps[-1]`` the data-space point after a full direct pass, in BOTH modes."""
        if self.training:
            grp = self.sibling_group()
            if grp is not None:
                res = grp.lists_for(self, p, g, mode)
                if res is not None:
                    return res
        return self.engine().run_lists(p, g, mode)

    def forward_fused(self, p, g, mode='inverse'):
        """-> (out, logdet): ``out`` = ps[0] (inverse) / ps[-1] (direct); ``logdet`` = sum_j logvars[j], (B,3,N)."""
        out, logdet, _ = self.engine().run(p, g, mode, False)
        return out, logdet

    def capture(self, p, g, mode='inverse', want_lists=False):
        """hipGraph-captured ``forward_fused`` bound to the storage of ``p`` and ``g``; see ``StackEngine.capture``.
        ``replay()`` returns ``[(out, logdet, lists)]``."""
        return self.engine().capture(p, g, mode, want_lists)


register_module_module_registration_hook(_note_parent)
