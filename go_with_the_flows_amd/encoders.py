"""Point-cloud encoder and the per-shape heads that sit immediately before the point-flow decoder.

Host-side mirror of lib/networks/encoders.py (PointNetCloudEncoder :9-28, FeatureEncoder :31-85, WeightsEncoder :87-91):
same constructors, attribute names and ``state_dict`` keys, so the reference's checkpoints load.

PointNetCloudEncoder in eval mode under ``torch.no_grad()`` runs the fused HIP kernel (csrc/gwtf_encoder.hip: all
SharedDot+BatchNorm+ReLU layers chained in registers on the MFMA, optional max-pool fused).  With batch-statistic
BatchNorm (``.train()``) or when a gradient is required it is a chain of plain library GEMMs and batch-norms on the HIP
device (torch.matmul -> rocBLAS, F.batch_norm -> MIOpen), which autograd differentiates.  CPU tensors raise: there is no
CPU path.  The per-shape heads are (B x in) x (in x out) library GEMMs.
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib
from ._lib import GwtfError, _ptr, _stream, check
from .layers import SharedDot, Swish


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

    def forward_max(self, input):
        """(B,3,N) -> (B,C_last): features max-pooled over points, what models.py:127-128 consumes; eval/no-grad never
        materialises the (B,C_last,N) tensor."""
        _ptr(input if input.is_contiguous() else input.contiguous(), 'input')
        if self._needs_graph(input):
            return torch.max(self.features(input), dim=2)[0]
        return self._fused(input, False, True)[1]


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

    def forward(self, input):
        h = self.features(input) if self.n_layers > 0 else input
        if self.deterministic:
            return self.mus(h)
        return self.mus(h), self.logvars(h)


class WeightsEncoder(FeatureEncoder):
    """Mixture-weight head: log-softmax of the deterministic output (reference encoders.py:87-91)."""

    def forward(self, input):
        return nn.functional.log_softmax(super().forward(input), dim=1)
