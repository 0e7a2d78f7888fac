"""Point-cloud decoder: a stack of ``n_flows`` coupling Triples run as ONE fused HIP launch.

Drop-in for the reference's ``LocalCondRNVPDecoder`` (lib/networks/decoders.py:41-79): same
constructor, attributes, ``state_dict`` keys, static ``get_param_count`` and
``forward(p, g, mode) -> (ps, mus, logvars)`` (direct-ordered lists of 3*n_flows ``(B,3,N)``
tensors).  ``forward_fused`` returns only what the reference's consumers read -- final
coordinates and the per-coordinate sum of logvars (losses.py:112-116, flow_mixture.py:175) --
and skips materialising the 9*n_flows intermediate tensors.
"""
import torch.nn as nn

from .flows import CondRealNVPFlow3DTriple, StackEngine


class LocalCondRNVPDecoder(nn.Module):
    def __init__(self, n_flows, f_n_features, g_n_features, weight_std=0.01):
        super().__init__()
        self.n_flows, self.f_n_features, self.g_n_features = n_flows, f_n_features, g_n_features
        self.weight_std = weight_std
        self.flows = nn.ModuleList(
            CondRealNVPFlow3DTriple(f_n_features, g_n_features, weight_std=weight_std, pattern=i % 2)
            for i in range(n_flows))
        self._engine = None

    @staticmethod
    def get_param_count(n_flows, f_n_features, g_n_features):
        """The reference's sizing formula (decoders.py:54-59); it under-counts the true module on purpose."""
        per_coupling = 18 * f_n_features + 4 * f_n_features * g_n_features + 6 * f_n_features ** 2
        return n_flows * 3 * per_coupling

    def engine(self):
        if self._engine is None:
            self._engine = StackEngine([c for t in self.flows for c in t.couplings()])
        return self._engine

    def forward(self, p, g, mode='direct'):
        """Reference contract (decoders.py:61-79).  ``ps[0]`` is the base-space point after a full
        inverse, ``ps[-1]`` the data-space point after a full direct pass, in BOTH modes."""
        return self.engine().run_lists(p, g, mode)

    def forward_fused(self, p, g, mode='inverse'):
        """-> (out, logdet): ``out`` = ps[0] (inverse) / ps[-1] (direct); ``logdet`` = sum_j logvars[j], (B,3,N)."""
        out, logdet, _ = self.engine().run(p, g, mode, False)
        return out, logdet

    def capture(self, p, g, mode='inverse', want_lists=False):
        """hipGraph-captured ``forward_fused`` bound to the storage of ``p`` and ``g``; see ``StackEngine.capture``.
        ``replay()`` returns ``[(out, logdet, lists)]``."""
        return self.engine().capture(p, g, mode, want_lists)
