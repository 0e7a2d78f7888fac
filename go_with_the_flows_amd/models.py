"""The callers on either side of the point-flow decoder: the VAE wrapper and the K-component flow-mixture model.

Host-side mirror of lib/networks/models.py:12-265 (Local_Cond_RNVP_MC_Global_RNVP_VAE) and
lib/networks/flow_mixture.py:11-179 (Flow_Mixture_Model) with lib/networks/losses.py:88-170 (FlowMixtureNLL,
Flow_Mixture_Loss): same constructor keywords, attribute names, ``state_dict`` keys (so the reference's checkpoints load
with strict=True), the same nested dict-of-lists outputs from ``forward`` and the same host-side randomness (torch RNG for
the reparameterisations, ``np.random.choice`` for the per-point component draw).

Besides the reference-shaped API there is the path a training loop should use: ``Flow_Mixture_Model.forward_fused`` runs
the encoder (fused HIP kernel in eval), the prior flow, ONE batched launch for the K decoders and returns the tensors
``Flow_Mixture_Loss.fused`` reduces with the mixture-NLL kernel -- no 9*n_flows lists, no B x K Python loop.

The image-conditioned variant (Flow_Mixture_SVR_Model, flow_mixture.py:181-239) is not mirrored: its ResNet-18 image
encoder is convolutional work outside the point-flow path (out of scope).
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from .decoders import LocalCondRNVPDecoder
from .encoders import FeatureEncoder, PointNetCloudEncoder, WeightsEncoder
from .mixture import MixtureStack, flow_mixture_nll
from .prior import GaussianEntropy, GaussianFlowNLL, GlobalRNVPDecoder


class Local_Cond_RNVP_MC_Global_RNVP_VAE(nn.Module):
    """Encoder + prior flow on the latent + ONE point-flow decoder (reference models.py:12-265)."""

    def __init__(self, **kwargs):
        super().__init__()
        g = kwargs.get
        self.train_mode, self.mode, self.deterministic = g('train_mode'), g('util_mode'), g('deterministic')
        self.pc_enc_init_n_channels = g('pc_enc_init_n_channels')
        self.pc_enc_init_n_features = g('pc_enc_init_n_features')
        self.pc_enc_n_features = g('pc_enc_n_features')
        self.g_latent_space_size = g('g_latent_space_size')
        self.g_prior_n_flows, self.g_prior_n_features = g('g_prior_n_flows'), g('g_prior_n_features')
        self.g_posterior_n_layers = g('g_posterior_n_layers')
        self.p_latent_space_size, self.p_prior_n_layers = g('p_latent_space_size'), g('p_prior_n_layers')
        self.p_decoder_n_flows, self.p_decoder_n_features = g('p_decoder_n_flows'), g('p_decoder_n_features')
        self.p_decoder_base_type, self.p_decoder_base_var = g('p_decoder_base_type'), g('p_decoder_base_var')

        self.pc_encoder = PointNetCloudEncoder(self.pc_enc_init_n_channels, self.pc_enc_init_n_features,
                                               self.pc_enc_n_features)
        G = self.g_latent_space_size
        self.g0_prior_mus = nn.Parameter(torch.empty(1, G).normal_(mean=0.0, std=0.033))         # models.py:66-70
        self.g0_prior_logvars = nn.Parameter(torch.empty(1, G).normal_(mean=0.0, std=0.33))
        self.g_prior = GlobalRNVPDecoder(self.g_prior_n_flows, self.g_prior_n_features, G, weight_std=0.01)
        self.g_posterior = FeatureEncoder(self.g_posterior_n_layers, self.pc_enc_n_features[-1], G, deterministic=False,
                                          mu_weight_std=0.0033, mu_bias=0.0, logvar_weight_std=0.033, logvar_bias=0.0)
        P = self.p_latent_space_size
        if self.p_decoder_base_type == 'free':
            self.p_prior = FeatureEncoder(self.p_prior_n_layers, G, P, deterministic=False, mu_weight_std=0.001,
                                          mu_bias=0.0, logvar_weight_std=0.01, logvar_bias=0.0)
        elif self.p_decoder_base_type == 'freevar':
            self.register_buffer('p_prior_mus', torch.zeros((1, P, 1)))
            self.p_prior = FeatureEncoder(self.p_prior_n_layers, G, P, deterministic=True, mu_weight_std=0.01, mu_bias=0.0)
        elif self.p_decoder_base_type == 'fixed':
            self.register_buffer('p_prior_mus', torch.zeros((1, P, 1)))
            self.register_buffer('p_prior_logvar', self.p_decoder_base_var * torch.ones((1, P, 1)))
        self.pc_decoder = self._build_pc_decoder(kwargs)

    def _build_pc_decoder(self, kwargs):
        return LocalCondRNVPDecoder(self.p_decoder_n_flows, self.p_decoder_n_features, self.g_latent_space_size,
                                    weight_std=0.01)

    def train(self, mode=True):
        """nn.Module.train + the data-parallel row-layout cache is dropped on a train <-> eval transition (an evaluation pass
        usually runs another per-rank batch size; dist.row_layout caches per-rank sizes by THIS rank's size only)."""
        if bool(mode) != self.training:
            from .dist import reset_row_layouts
            reset_row_layouts()
        return super().train(mode)

    def reparameterize(self, mu, logvar):
        """mu + exp(0.5 logvar) * N(0,1)   (models.py:99-109; torch RNG of the tensors' device)."""
        std = torch.exp(0.5 * logvar)
        return torch.randn_like(std).mul(std).add_(mu)

    def _pooled_features(self, g_input):
        return self.pc_encoder.forward_max(g_input)          # == torch.max(self.pc_encoder(g_input), dim=2)[0]

    def encode(self, g_input, defer_prior=False):
        """models.py:111-151: posterior from the cloud (training / autoencoding) or a draw from the learned prior
        (generating), pushed through the prior flow; lists are ordered base -> data.
        defer_prior=True (training mode): the prior flow is launched on a side stream and its lists are filled in by
        ``finish_encode(out)`` -- call it after the decoders have been launched (forward_fused does)."""
        B, G = g_input.shape[0], self.g_latent_space_size
        out = {'g_prior_mus': [self.g0_prior_mus.expand(B, G)], 'g_prior_logvars': [self.g0_prior_logvars.expand(B, G)]}
        out['_g0_params'] = (self.g0_prior_mus, self.g0_prior_logvars)        # (the unexpanded base Gaussian, for the fused loss)
        if self.mode in ('training', 'autoencoding'):
            out['g_posterior_mus'], out['g_posterior_logvars'] = self.g_posterior(self._pooled_features(g_input))
            out['g_posterior_samples'] = (self.reparameterize(out['g_posterior_mus'], out['g_posterior_logvars'])
                                          if self.mode == 'training' else out['g_posterior_mus'])
            if defer_prior and hasattr(self.g_prior, 'forward_async'):
                out['_prior_handle'] = self.g_prior.forward_async(out['g_posterior_samples'], mode='inverse')
                return out
            buf_g = self.g_prior(out['g_posterior_samples'], mode='inverse')
            from .prior import stacked_lists
            stacked = stacked_lists(self.g_prior)
            if stacked is not None:
                out['_g_prior_logvars_stacked'] = stacked[2]
            out['g_prior_samples'] = buf_g[0] + [out['g_posterior_samples']]
        elif self.mode == 'generating':
            out['g_prior_samples'] = [self.reparameterize(out['g_prior_mus'][0], out['g_prior_logvars'][0])]
            buf_g = self.g_prior(out['g_prior_samples'][0], mode='direct')
            out['g_prior_samples'] += buf_g[0]
        else:
            raise ValueError(f'unknown util_mode {self.mode!r}')
        out['g_prior_mus'] += buf_g[1]
        out['g_prior_logvars'] += buf_g[2]
        return out

    def finish_encode(self, out):
        """Join the deferred prior flow (see encode(defer_prior=True)) and complete the three prior lists."""
        handle = out.pop('_prior_handle', None)
        if handle is not None:
            buf_g = handle.result()
            from .prior import stacked_lists
            stacked = stacked_lists(self.g_prior)
            if stacked is not None:
                out['_g_prior_logvars_stacked'] = stacked[2]
            out['g_prior_samples'] = buf_g[0] + [out['g_posterior_samples']]
            out['g_prior_mus'] += buf_g[1]
            out['g_prior_logvars'] += buf_g[2]
        return out

    def _base_gaussian(self, g_sample):
        """(mu0, lv0) of the decoder's base distribution, each broadcastable to (B,P,1)  (models.py:169-193)."""
        if self.p_decoder_base_type == 'free':
            mu0, lv0 = self.p_prior(g_sample)
            return mu0.unsqueeze(2), lv0.unsqueeze(2)
        if self.p_decoder_base_type == 'freevar':
            return self.p_prior_mus, self.p_prior(g_sample).unsqueeze(2)
        if self.p_decoder_base_type == 'fixed':
            return self.p_prior_mus, self.p_prior_logvar
        raise ValueError(f'unknown p_decoder_base_type {self.p_decoder_base_type!r}')

    def _base_gaussian_repeated(self, g_sample, times):
        """_base_gaussian evaluated ONCE with the buffer side effects of `times` evaluations on the same batch (the reference
        evaluates p_prior inside its loop over the K components, models.py:169-193 in flow_mixture.py:163-166): every tracked
        BatchNorm of p_prior in train mode gets running = (1-m) running + m batch applied `times` times and num_batches_tracked
        += times (FeatureEncoder.forward(bn_updates=times))."""
        if times <= 1 or self.p_decoder_base_type == 'fixed':
            return self._base_gaussian(g_sample)
        if self.p_decoder_base_type == 'free':
            mu0, lv0 = self.p_prior(g_sample, bn_updates=times)
            return mu0.unsqueeze(2), lv0.unsqueeze(2)
        return self.p_prior_mus, self.p_prior(g_sample, bn_updates=times).unsqueeze(2)

    def one_flow_decode(self, p_input, g_sample, pc_decoder, n_sampled_points):
        """models.py:153-207: lists for ONE component (inverse on p_input when training, direct on a base draw otherwise)."""
        B, P = g_sample.shape[0], self.p_latent_space_size
        mu0, lv0 = self._base_gaussian(g_sample)
        out = {'p_prior_mus': [mu0.expand(B, P, n_sampled_points)], 'p_prior_logvars': [lv0.expand(B, P, n_sampled_points)]}
        if self.mode == 'training':
            buf = pc_decoder(p_input, g_sample, mode='inverse')
            out['p_prior_samples'] = buf[0] + [p_input]
        else:
            out['p_prior_samples'] = [self.reparameterize(out['p_prior_mus'][0], out['p_prior_logvars'][0])]
            buf = pc_decoder(out['p_prior_samples'][0], g_sample, mode='direct')
            out['p_prior_samples'] += buf[0]
        out['p_prior_mus'] += buf[1]
        out['p_prior_logvars'] += buf[2]
        return out

    def decode(self, p_input, g_sample, n_sampled_points, labeled_samples=False, warmup=False):
        """Single-decoder decode (the body the reference keeps commented out at models.py:209-223)."""
        return self.one_flow_decode(p_input, g_sample, self.pc_decoder, n_sampled_points), None

    def forward(self, g_input, p_input, images=None, n_sampled_points=None, labeled_samples=False, warmup=False):
        """models.py:224-265."""
        size = p_input.shape[2] if n_sampled_points is None else n_sampled_points
        if images is not None and self.train_mode == 'p_rnvp_mc_g_rnvp_vae_ic':
            raise NotImplementedError('image-conditioned encoding (ResNet-18) is outside the point-flow path')
        # training: the prior flow (a latency chain on one compute unit) runs on a side stream beside the decoders, as in
        # forward_fused; its lists are joined in before the outputs are handed back
        output_encoder = self.encode(g_input, defer_prior=self.mode == 'training')
        g_sample = (output_encoder['g_posterior_samples'] if self.mode in ('training', 'autoencoding')
                    else output_encoder['g_prior_samples'][-1])
        if labeled_samples:
            samples, labels, logits = self.decode(p_input, g_sample, size, labeled_samples, warmup)
            return self.finish_encode(output_encoder), samples, labels, logits
        output_decoder, logits = self.decode(p_input, g_sample, size, labeled_samples, warmup)
        return self.finish_encode(output_encoder), output_decoder, logits


class Flow_Mixture_Model(Local_Cond_RNVP_MC_Global_RNVP_VAE):
    """K point-flow decoders with per-shape mixture weights (reference flow_mixture.py:11-179)."""

    def __init__(self, **kwargs):
        self.n_components = kwargs['n_components']
        self.params_reduce_mode = kwargs['params_reduce_mode']
        self.weights_type = kwargs['weights_type']
        super().__init__(**kwargs)
        # registered after the base class's parameters, as in the reference (state_dict order)
        self.mixture_weights_logits = nn.Parameter(torch.zeros(self.n_components), requires_grad=True)
        self.mixture_weights_encoder = WeightsEncoder(3, self.g_latent_space_size, self.n_components, deterministic=True,
                                                      mu_weight_std=0.001, mu_bias=0.0, logvar_weight_std=0.01,
                                                      logvar_bias=0.0)
        self._stack = None

    def _build_pc_decoder(self, kwargs):
        n_flows, f = self._get_decoder_params()
        return nn.ModuleList([LocalCondRNVPDecoder(n_flows, f, self.g_latent_space_size, weight_std=0.01)
                              for _ in range(self.n_components)])

    # -- parameter-budget rules (flow_mixture.py:44-102) --------------------------------------------------------------
    def _get_decoder_params(self):
        n = self.n_components
        if n == 1 or self.params_reduce_mode == 'none':
            return self.p_decoder_n_flows, self.p_decoder_n_features
        count = LocalCondRNVPDecoder.get_param_count
        if self.params_reduce_mode == 'depth_and_feature':
            depth = math.ceil(self.p_decoder_n_flows / math.sqrt(n))
            f, _ = self._get_p_decoder_n_features(depth)
        elif self.params_reduce_mode == 'depth_first':
            depth = math.ceil(self.p_decoder_n_flows / n)
            f, _ = self._get_p_decoder_n_features(depth)
        elif self.params_reduce_mode == 'feature_first':
            depth = self.p_decoder_n_flows
            f, (over, budget, total) = self._get_p_decoder_n_features(depth)
            if over:
                while total > budget:
                    depth -= 1
                    total = count(depth, f, self.g_latent_space_size) * n
        else:
            raise ValueError(f'Unknown params_reduce_mode: {self.params_reduce_mode}')
        return depth, f

    def _get_p_decoder_n_features(self, depth):
        f, n, G = self.p_decoder_n_features, self.n_components, self.g_latent_space_size
        count = LocalCondRNVPDecoder.get_param_count
        budget = count(self.p_decoder_n_flows, f, G)
        total = budget * n
        while total > budget and f > 4:
            f -= 1
            total = count(depth, f, G) * n
        return f, (total > budget, budget, total)

    # -- mixture weights and decoding (flow_mixture.py:104-179) ------------------------------------------------------
    def get_weights(self, g_sample, warmup=False):
        if warmup or self.weights_type == 'global_weights':
            return self.mixture_weights_logits.unsqueeze(0).expand(g_sample.shape[0], self.n_components)
        if self.weights_type == 'learned_weights':
            return self.mixture_weights_encoder(g_sample)
        raise ValueError(f'unknown weights_type {self.weights_type!r}')

    def _draw_components(self, logits_row, n_points):
        """np.random.choice over the normalised weights of ONE shape (flow_mixture.py:146-160) -> per-point component."""
        w = np.exp(logits_row.detach().cpu().numpy())
        return np.random.choice(range(self.n_components), size=n_points, p=w / w.sum())

    def decode(self, p_input, g_sample, n_sampled_points, labeled_samples=False, warmup=False):
        logits = self.get_weights(g_sample, warmup)
        K = self.n_components
        if self.mode == 'training':
            sizes = [n_sampled_points] * K
        else:
            assert p_input.shape[0] == 1                     # evaluation feeds one shape at a time (flow_mixture.py:146)
            flows_idx = self._draw_components(logits[0], n_sampled_points)
            sizes = [int((flows_idx == t).sum()) for t in range(K)]
        # literal_k_loop: this mirror's own shortcut off -- the K decoders are called one at a time exactly as the reference's loop
        # does (flow_mixture.py:163-166), which is what a maintainer who only swaps the decoder import runs (tools/bench_train.py
        # --api swap; the decoders' sibling batching, decoders._SiblingGroup, then does the batching below the API)
        batched = self.mode == 'training' and not getattr(self, 'literal_k_loop', False)
        output_decoder = self._decode_training_batched(p_input, g_sample, sizes) if batched else None
        if output_decoder is None:
            output_decoder = [self.one_flow_decode(p_input, g_sample, self.pc_decoder[i], sizes[i]) for i in range(K)]
        if not labeled_samples:
            return output_decoder, logits
        samples = torch.zeros_like(p_input)
        labels = torch.zeros(p_input.size(0), p_input.size(2))
        for t in range(K):
            mask = torch.from_numpy(flows_idx == t)
            samples[:, :, mask] = output_decoder[t]['p_prior_samples'][-1]
            labels[:, mask] = t + 1
        return samples, labels, logits

    def _decode_training_batched(self, p_input, g_sample, sizes):
        """The K one_flow_decode calls of the training pass (flow_mixture.py:163-166) as ONE pass of the K-batched train pipeline:
        the same K dicts of lists (entries are slices of stacked tensors).  None when it does not apply (eval-mode BatchNorm, CPU
        tensors, sub-sampled clouds): the per-decoder calls then run."""
        if not p_input.is_cuda or any(sz != p_input.shape[2] for sz in sizes) or not self.pc_decoder[0].training:
            return None
        res = self.mixture_stack().forward_all_lists(p_input, g_sample, mode='inverse')
        if res is None:
            return None
        z, logdet, (ps, mus, lvs) = res                          # (K,B,3,N), (K,B,3,N), (K,C,B,3,N) each
        B, P, n = g_sample.shape[0], self.p_latent_space_size, p_input.shape[2]
        mu0, lv0 = self._base_gaussian_repeated(g_sample, self.n_components)
        out = []
        for k in range(self.n_components):
            # slot 0 of the samples IS the stack's output and sum(logvars[1:]) IS its log-det: handing the loss those two tensors
            # (the list entries hold the same values) keeps the gradient off the per-slot route, whose backward would materialise
            # a (K, C, B, 3, N) gradient tensor per list
            out.append({'p_prior_mus': [mu0.expand(B, P, n)] + list(mus[k].unbind(0)),
                        'p_prior_logvars': [lv0.expand(B, P, n)] + list(lvs[k].unbind(0)),
                        'p_prior_samples': [z[k]] + list(ps[k].unbind(0))[1:] + [p_input],
                        '_sum_flow_logvars': logdet[k]})
        return out

    # -- the fused path ------------------------------------------------------------------------------------------------
    def mixture_stack(self):
        if self._stack is None:
            self._stack = MixtureStack(self.pc_decoder)
        return self._stack

    def forward_fused(self, g_input, p_input, warmup=False):
        """Training / density pass without the list API: -> (output_encoder, dec) with dec = {z, logdet (K,B,3,N); mu0, lv0
        (K,B,P); logits (B,K)}: everything ``Flow_Mixture_Loss.fused`` needs.  Same arithmetic as ``forward`` in 'training'
        mode (one encoder pass, the prior flow, every component on every point in ONE batched launch when BatchNorm is in
        eval mode and no gradient is needed; per component otherwise)."""
        if self.mode != 'training':
            raise ValueError("forward_fused is the density ('training') pass; use sample_fused for generation")
        output_encoder = self.encode(g_input, defer_prior=True)     # the prior flow runs beside the decoders (side stream)
        g_sample = output_encoder['g_posterior_samples']
        logits = self.get_weights(g_sample, warmup)
        B, K, P = g_sample.shape[0], self.n_components, self.p_latent_space_size
        # The reference evaluates p_prior once per component (models.py:169-193 inside the K loop of flow_mixture.py:163-166): K
        # identical outputs, and K running-statistic updates with the same batch statistics.  One evaluation here; its BatchNorm
        # modules replay the other K - 1 buffer updates (same values as K passes would leave, to rounding).
        m, v = self._base_gaussian_repeated(g_sample, K)
        # (K views of the one evaluation: no copy here, and ONE sum over K in the backward where a stack's would be K - 1 adds)
        mu0 = m.expand(B, P, 1)[:, :, 0].unsqueeze(0).expand(K, B, P)
        lv0 = v.expand(B, P, 1)[:, :, 0].unsqueeze(0).expand(K, B, P)
        z, logdet = self.mixture_stack().forward_all(p_input, g_sample, mode='inverse')
        self.finish_encode(output_encoder)
        return output_encoder, {'z': z, 'logdet': logdet, 'mu0': mu0, 'lv0': lv0, 'logits': logits}

    @torch.no_grad()
    def sample_fused(self, g_sample, n_points, return_labels=False):
        """Generation for ONE shape (flow_mixture.py:146-177 without the K Python passes): draw each point's component,
        draw base samples, push every point through its own component in one launch.  -> (1,3,n_points)[, labels]."""
        assert g_sample.shape[0] == 1
        logits = self.get_weights(g_sample)
        flows_idx = np.sort(self._draw_components(logits[0], n_points))       # points laid out component by component
        counts = [int((flows_idx == t).sum()) for t in range(self.n_components)]
        P = self.p_latent_space_size
        mu0, lv0 = self._base_gaussian(g_sample)
        z0 = self.reparameterize(mu0.expand(1, P, n_points), lv0.expand(1, P, n_points))
        x, _ = self.mixture_stack().forward_partition(z0, g_sample, counts, mode='direct')
        if return_labels:
            return x, torch.from_numpy(flows_idx + 1).to(x.device).unsqueeze(0).float()
        return x

    @staticmethod
    def _padded_partition(labels, K):
        """labels (S, n) ints in [0, K): every sample's per-point component.  -> counts (K,) = the LARGEST count of each component
        over the samples, and index tensors: slot (S, n) = where point i of sample s sits in the padded component-major layout
        (points of component k of every sample occupy [start_k, start_k + counts_k), the sample's own in front)."""
        S, n = labels.shape
        per = np.stack([(labels == k).sum(1) for k in range(K)], axis=1)            # (S, K)
        counts = per.max(0)
        starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
        slot = np.empty((S, n), np.int64)
        for s in range(S):
            order = np.argsort(labels[s], kind='stable')                                # points of component 0 first, ...
            rank_in_comp = np.concatenate([np.arange(c) for c in per[s]]) if n else np.zeros(0, np.int64)
            slot[s, order] = starts[labels[s][order]] + rank_in_comp
        return counts, slot

    @torch.no_grad()
    def sample_many(self, g_samples, n_points, return_labels=False):
        """Generation for S shapes in ONE launch (the reference samples one shape per call, flow_mixture.py:146-177; the decoder
        launch for a single 2048-point shape is latency-bound at ~36 us -- bench.py k16_b1 -- where a batch runs 22 x that rate).
        Each sample draws its points' components from its own mixture weights (np.random.choice, sample by sample, as the
        reference's loop would); the points are laid out component by component with every component's segment padded to its
        largest count over the batch (a few per cent of padding: multinomial counts concentrate), pushed through their own
        component by one partitioned launch, and gathered back in the points' original order.
        -> (S, 3, n_points)[, labels (S, n_points) in 1..K]."""
        S, K, P = g_samples.shape[0], self.n_components, self.p_latent_space_size
        logits = self.get_weights(g_samples)
        labels = np.stack([self._draw_components(logits[s], n_points) for s in range(S)])
        mu0, lv0 = self._base_gaussian(g_samples)
        z0 = self.reparameterize(mu0.expand(S, P, n_points), lv0.expand(S, P, n_points))
        x = self._decode_partitioned_many(z0, g_samples, labels)
        if return_labels:
            return x, torch.from_numpy(labels + 1).to(x.device).float()
        return x

    def _decode_partitioned_many(self, z0, g_samples, labels):
        """z0 (S, 3, n) base samples, labels (S, n) numpy: point i of sample s through component labels[s, i] -> (S, 3, n)."""
        S, _, n = z0.shape
        counts, slot = self._padded_partition(np.asarray(labels), self.n_components)
        n_pad = int(counts.sum())
        slot_t = torch.from_numpy(slot).to(z0.device)
        zp = torch.zeros(S, 3, n_pad, device=z0.device, dtype=torch.float32)
        zp.scatter_(2, slot_t.unsqueeze(1).expand(S, 3, n), z0.float())
        xp, _ = self.mixture_stack().forward_partition(zp, g_samples, [int(c) for c in counts], mode='direct')
        return xp.gather(2, slot_t.unsqueeze(1).expand(S, 3, n))


def _restack(tensors):
    """torch.stack(tensors) -- or, when they are exactly the K slices `base[0], ..., base[K-1]` of one (K, ...) tensor in order (what a
    batched decode hands out), that tensor itself: no copy forward, no unbind / stack pair backward."""
    base = tensors[0]._base
    if (base is not None and base.dim() == tensors[0].dim() + 1 and base.shape[0] == len(tensors) and base.is_contiguous()
            and base.requires_grad == tensors[0].requires_grad):
        step = base.stride(0) * base.element_size()
        if all(t._base is base and t.shape == base.shape[1:] and t.is_contiguous() and t.data_ptr() == base.data_ptr() + k * step
               for k, t in enumerate(tensors)):
            return base
    return torch.stack(tensors)


def _first_column(t):
    """t[:, :, 0] of a (B, P, n) base-Gaussian entry.  The reference builds that entry as `head(g).unsqueeze(2).expand(B, P, n)`
    (models.py:169-193): when `t` is such a broadcast of a (B, P) tensor, that tensor itself is returned -- same values, and the
    gradient reaches it directly instead of through a zero-filled (B, P, n) tensor and a sum over its n columns per component."""
    base = t._base
    if (base is not None and t.dim() == 3 and base.dim() == 2 and t.stride(2) == 0 and base.shape == t.shape[:2]
            and base.stride() == t.stride()[:2] and base.data_ptr() == t.data_ptr()):
        return base
    return t[:, :, 0]


class FlowMixtureNLL(nn.Module):
    """Mixture point NLL on the reference's list outputs (losses.py:88-137), reduced by the fused HIP kernel."""

    def per_shape(self, output_decoder, mixture_weights_logits):
        """-> (B,) NLL of every shape (its mean is the reference's scalar)."""
        z = _restack([o['p_prior_samples'][0] for o in output_decoder])
        # the batched training decode attaches the stack's own log-det (= sum of the list's flow entries, same values)
        def flow_logdet(o):
            if '_sum_flow_logvars' in o:
                return o['_sum_flow_logvars']
            from .decoders import slot_sum
            own = slot_sum(o['p_prior_logvars'][1:])         # the K-loop's lists from a batched sibling round: the sum exists already
            return own if own is not None else sum(o['p_prior_logvars'][1:])
        logdet = _restack([flow_logdet(o) for o in output_decoder])
        mu0 = torch.stack([_first_column(o['p_prior_mus'][0]) for o in output_decoder])
        lv0 = torch.stack([_first_column(o['p_prior_logvars'][0]) for o in output_decoder])
        return flow_mixture_nll(z, logdet, mu0, lv0, mixture_weights_logits)[1]

    def forward(self, output_decoder, mixture_weights_logits):
        return self.per_shape(output_decoder, mixture_weights_logits).mean()


class Flow_Mixture_Loss(nn.Module):
    """pnll_weight * pnll + gnll_weight * gnll - gent_weight * gent   (losses.py:140-170)."""

    def __init__(self, **kwargs):
        super().__init__()
        self.pnll_weight, self.gnll_weight, self.gent_weight = kwargs.get('pnll_weight'), kwargs.get('gnll_weight'), kwargs.get('gent_weight')
        self.n_components = kwargs.get('n_components')
        self.PNLL, self.GNLL, self.GENT = FlowMixtureNLL(), GaussianFlowNLL(), GaussianEntropy()

    def _combine(self, nll, output_prior):
        """nll: (B,) per-shape point NLL (or its mean: the three terms are then combined with torch ops)."""
        fused = self._combine_fused(nll, output_prior) if nll.dim() == 1 else None
        if fused is not None:
            return fused
        pnll = nll.mean() if nll.dim() == 1 else nll
        gnll = self.GNLL(output_prior['g_prior_samples'], output_prior['g_prior_mus'], output_prior['g_prior_logvars'],
                         output_prior.get('_g_prior_logvars_stacked'))
        gent = self.GENT(output_prior['g_posterior_logvars'])
        return self.pnll_weight * pnll + self.gnll_weight * gnll - self.gent_weight * gent, pnll, gnll, gent

    def _combine_fused(self, nll, output_prior):
        """All four values in one launch (prior.LatentLossFn) when the prior flow handed its logvars over as one tensor."""
        flow_lv, g0 = output_prior.get('_g_prior_logvars_stacked'), output_prior.get('_g0_params')
        post_lv, z = output_prior.get('g_posterior_logvars'), output_prior['g_prior_samples'][0]
        if flow_lv is None or g0 is None or post_lv is None or os.environ.get('GWTF_NO_FUSED_LATENT_LOSS') == '1':
            return None
        ts = (nll, z, g0[0], g0[1], flow_lv, post_lv)
        if not all(t.is_cuda and t.dtype == torch.float32 for t in ts) or z.dim() != 2:
            return None
        from .prior import LatentLossFn
        vals = LatentLossFn.apply(nll.contiguous(), z.contiguous(), g0[0].reshape(-1), g0[1].reshape(-1), flow_lv.contiguous(),
                                  post_lv.contiguous(), float(self.pnll_weight), float(self.gnll_weight), float(self.gent_weight))
        return vals[0], vals[1], vals[2], vals[3]

    def forward(self, output_prior, output_decoder, mixture_weights_logits):
        return self._combine(self.PNLL.per_shape(output_decoder, mixture_weights_logits), output_prior)

    def fused(self, output_prior, dec):
        """Same four values from ``Flow_Mixture_Model.forward_fused``'s outputs."""
        return self._combine(flow_mixture_nll(dec['z'], dec['logdet'], dec['mu0'], dec['lv0'], dec['logits'])[1], output_prior)
