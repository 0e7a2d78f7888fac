#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the GENUINE reference on CPU.

Build-container only: needs /root/reference (read-only).  The reference is imported, never
copied; only inputs/outputs (data) are written.  Weights are not stored: they are regenerated from
a seed by ``go_with_the_flows_amd.synth.synth_state`` (pure numpy), and this script asserts that the
reference's ``state_dict`` keys/shapes/dtypes equal our module's, which pins the checkpoint contract.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz and contract.json
"""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
if not os.path.isdir(REF):
    sys.exit('reference checkout not present: fixtures can only be regenerated in the build container')
sys.path.insert(0, ROOT)
sys.path.insert(1, REF)

import numpy as np
import torch

from lib.networks import flows as rflows            # noqa: E402  (the reference)
from lib.networks import decoders as rdec           # noqa: E402
from lib.networks import losses as rloss            # noqa: E402
from lib.networks.flow_mixture import Flow_Mixture_Model  # noqa: E402
import go_with_the_flows_amd as ours                # noqa: E402
from go_with_the_flows_amd.synth import synth_state, synth_inputs  # noqa: E402

torch.set_num_threads(4)
T = torch.from_numpy


def npy(x):
    return x.detach().cpu().numpy().copy()      # copy: .numpy() aliases module buffers that later runs overwrite


def load_into(ref_module, our_module, seed):
    rs, os_ = ref_module.state_dict(), our_module.state_dict()
    assert list(rs.keys()) == list(os_.keys()), 'state_dict key order/names differ from the reference'
    for k in rs:
        assert rs[k].shape == os_[k].shape and rs[k].dtype == os_[k].dtype, k
    st = synth_state(os_, seed)
    ref_module.load_state_dict({k: T(v) for k, v in st.items()})
    return st


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrays)
    print(f'{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB')


BN_PROBES = ('T_mu_0.mu_sd0_bn', 'T_logvar_0.logvar_sd1_bn', 'T_mu_0_cond_w.mu_sd1_film_w0_bn',
             'T_logvar_0_cond_b.logvar_sd1_film_b0_bn')


def g1_single_couplings():
    """Every warp pattern, both modes, eval and train BatchNorm."""
    f, G, B, N = 8, 16, 3, 16
    out = {'dims': np.array([f, G, B, N])}
    for pi, warp in enumerate(ours.WARP_PATTERNS):
        ref = rflows.CondRealNVPFlow3D(f, G, warp_inds=list(warp))
        mine = ours.CondRealNVPFlow3D(f, G, warp_inds=list(warp))
        seed = 100 + pi
        p, g = synth_inputs(B, N, G, 200 + pi)
        out[f'p{pi}'], out[f'g{pi}'] = p, g
        for training in (False, True):
            for mode in ('direct', 'inverse'):
                load_into(ref, mine, seed)
                ref.train(training)
                with torch.no_grad():
                    po, mu, lv = ref(T(p), T(g), mode=mode)
                tag = f'{pi}_{"train" if training else "eval"}_{mode}'
                out['pout_' + tag], out['mu_' + tag], out['lv_' + tag] = npy(po), npy(mu), npy(lv)
                if training:
                    sd = ref.state_dict()
                    for probe in BN_PROBES:
                        out[f'rm_{tag}_{probe}'] = npy(sd[probe + '.running_mean'])
                        out[f'rv_{tag}_{probe}'] = npy(sd[probe + '.running_var'])
    save('g1_couplings', **out)


def g2_triples():
    f, G, B, N = 8, 16, 2, 16
    out = {'dims': np.array([f, G, B, N])}
    for pattern in (0, 1):
        ref = rflows.CondRealNVPFlow3DTriple(f, G, pattern=pattern)
        mine = ours.CondRealNVPFlow3DTriple(f, G, pattern=pattern)
        load_into(ref, mine, 300 + pattern)
        ref.eval()
        p, g = synth_inputs(B, N, G, 310 + pattern)
        out[f'p{pattern}'], out[f'g{pattern}'] = p, g
        for mode in ('direct', 'inverse'):
            with torch.no_grad():
                ps, mus, lvs = ref(T(p), T(g), mode=mode)
            out[f'ps_{pattern}_{mode}'] = np.stack([npy(t) for t in ps])
            out[f'mus_{pattern}_{mode}'] = np.stack([npy(t) for t in mus])
            out[f'lvs_{pattern}_{mode}'] = np.stack([npy(t) for t in lvs])
    save('g2_triples', **out)


def decoder_case(name, L, f, G, B, N, seed, full_lists, train_too):
    ref = rdec.LocalCondRNVPDecoder(L, f, G)
    mine = ours.LocalCondRNVPDecoder(L, f, G)
    p, g = synth_inputs(B, N, G, seed + 1)
    out = {'dims': np.array([L, f, G, B, N, seed]), 'p': p, 'g': g,
           'param_count': np.array(rdec.LocalCondRNVPDecoder.get_param_count(L, f, G))}
    for training in ((False, True) if train_too else (False,)):
        for mode in ('direct', 'inverse'):
            load_into(ref, mine, seed)
            ref.train(training)
            with torch.no_grad():
                ps, mus, lvs = ref(T(p), T(g), mode=mode)
            tag = f'{"train" if training else "eval"}_{mode}'
            out['first_' + tag], out['last_' + tag] = npy(ps[0]), npy(ps[-1])
            out['logdet_' + tag] = npy(sum(lvs))
            if full_lists:
                out['ps_' + tag] = np.stack([npy(t) for t in ps])
                out['mus_' + tag] = np.stack([npy(t) for t in mus])
                out['lvs_' + tag] = np.stack([npy(t) for t in lvs])
            # fp64 run of the same reference module: sizes the tolerance
            if not training:
                ref64 = rdec.LocalCondRNVPDecoder(L, f, G).double()
                ref64.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in ref.state_dict().items()})
                ref64.eval()
                with torch.no_grad():
                    ps64, _, lvs64 = ref64(T(p).double(), T(g).double(), mode=mode)
                out['first64_' + tag] = npy(ps64[0])
                out['last64_' + tag] = npy(ps64[-1])
                out['logdet64_' + tag] = npy(sum(lvs64))
    save(name, **out)


def g5_losses():
    """PointFlowNLL (K=1) and FlowMixtureNLL (K=4, non-uniform logits) on genuine decoder outputs."""
    L, f, G, B, N, K = 2, 8, 16, 3, 32, 4
    out = {'dims': np.array([L, f, G, B, N, K])}
    p, g = synth_inputs(B, N, G, 501)
    rng = np.random.default_rng(502)
    logits = rng.normal(0, 1.0, (B, K)).astype(np.float32)
    mu0 = rng.normal(0, 0.2, (K, B, 3)).astype(np.float32)
    lv0 = rng.normal(-0.5, 0.3, (K, B, 3)).astype(np.float32)
    out.update(p=p, g=g, logits=logits, mu0=mu0, lv0=lv0)
    comps, zs, lds = [], [], []
    for k in range(K):
        ref = rdec.LocalCondRNVPDecoder(L, f, G)
        mine = ours.LocalCondRNVPDecoder(L, f, G)
        load_into(ref, mine, 510 + k)
        ref.eval()
        with torch.no_grad():
            ps, mus, lvs = ref(T(p), T(g), mode='inverse')
        m0 = T(mu0[k]).unsqueeze(2).expand(B, 3, N)
        l0 = T(lv0[k]).unsqueeze(2).expand(B, 3, N)
        # exactly how one_flow_decode assembles the lists (reference models.py:195-205)
        comps.append({'p_prior_samples': ps + [T(p)], 'p_prior_mus': [m0] + mus, 'p_prior_logvars': [l0] + lvs})
        zs.append(npy(ps[0]))
        lds.append(npy(sum(lvs)))
    out['z'], out['logdet'] = np.stack(zs), np.stack(lds)
    with torch.no_grad():
        out['pointflow_nll_k0'] = npy(rloss.PointFlowNLL()(comps[0]))
        out['mixture_nll'] = npy(rloss.FlowMixtureNLL()(comps, T(logits)))
        out['mixture_nll_k1'] = npy(rloss.FlowMixtureNLL()(comps[:1], T(logits[:, :1])))
    save('g5_losses', **out)


def g6_keep_drift():
    """33 couplings with zeroed output layers: every coordinate drifts by sqrt(1+eps)^33."""
    L, f, G, B, N = 11, 8, 16, 1, 8
    ref = rdec.LocalCondRNVPDecoder(L, f, G)
    mine = ours.LocalCondRNVPDecoder(L, f, G)
    st = load_into(ref, mine, 600)
    sd = ref.state_dict()
    for k in sd:
        if k.endswith('sd2.weight') or k.endswith('sd2.bias'):
            sd[k].zero_()
    ref.load_state_dict(sd)
    ref.eval()
    p, g = synth_inputs(B, N, G, 601)
    out = {'dims': np.array([L, f, G, B, N]), 'p': p, 'g': g}
    for mode in ('direct', 'inverse'):
        with torch.no_grad():
            ps, _, lvs = ref(T(p), T(g), mode=mode)
        out['out_' + mode] = npy(ps[-1] if mode == 'direct' else ps[0])
        out['logdet_' + mode] = npy(sum(lvs))
    save('g6_keep_drift', **out)


def g7_model_training_forward():
    """Caller semantics: full Flow_Mixture_Model.forward in training mode + Flow_Mixture_Loss (tiny dims).
    Captures what crosses the decoder boundary (g_sample, base Gaussians, logits) and the loss terms."""
    cfg = dict(train_mode='p_rnvp_mc_g_rnvp_vae', util_mode='training', deterministic=False,
               pc_enc_init_n_channels=3, pc_enc_init_n_features=8, pc_enc_n_features=[16, 32],
               g_latent_space_size=16, g_prior_n_flows=2, g_prior_n_features=16, g_posterior_n_layers=1,
               p_latent_space_size=3, p_prior_n_layers=1, p_decoder_n_flows=2, p_decoder_n_features=8,
               p_decoder_base_type='free', p_decoder_base_var=-3.9551, n_components=2,
               params_reduce_mode='none', weights_type='learned_weights',
               pnll_weight=1.0, gnll_weight=1.0, gent_weight=1.0)
    torch.manual_seed(7)
    model = Flow_Mixture_Model(**cfg)
    K = cfg['n_components']
    decoder_states = []
    for k in range(K):
        mine = ours.LocalCondRNVPDecoder(2, 8, 16)
        decoder_states.append(load_into(model.pc_decoder[k], mine, 700 + k))
    model.eval()  # BatchNorm in eval; `mode` attribute stays 'training' so the inverse path is taken
    B, N = 3, 32
    p, _ = synth_inputs(B, N, 16, 710)
    gcloud, _ = synth_inputs(B, N, 16, 711)
    captured = {}

    def hook(mod, args, kwargs, output):
        captured.setdefault('g_sample', npy(args[1]))

    h = model.pc_decoder[0].register_forward_hook(hook, with_kwargs=True)
    torch.manual_seed(8)
    with torch.no_grad():
        out_enc, out_dec, logits = model(T(gcloud), T(p), None, None, False, False)
        loss, pnll, gnll, gent = rloss.Flow_Mixture_Loss(**cfg)(out_enc, out_dec, logits)
    h.remove()
    save('g7_model_forward',
         dims=np.array([B, N, K]), p=p, g_sample=captured['g_sample'], logits=npy(logits),
         mu0=np.stack([npy(o['p_prior_mus'][0][:, :, 0]) for o in out_dec]),
         lv0=np.stack([npy(o['p_prior_logvars'][0][:, :, 0]) for o in out_dec]),
         z=np.stack([npy(o['p_prior_samples'][0]) for o in out_dec]),
         logdet=np.stack([npy(sum(o['p_prior_logvars'][1:])) for o in out_dec]),
         n_lists=np.array([len(out_dec[0]['p_prior_samples']), len(out_dec[0]['p_prior_mus']),
                           len(out_dec[0]['p_prior_logvars'])]),
         loss=npy(loss), pnll=npy(pnll), gnll=npy(gnll), gent=npy(gent))


def g8_gradients():
    """Gradients of a scalar NLL-like loss through a small decoder (inverse), for the backward kernels."""
    L, f, G, B, N = 2, 8, 16, 2, 32
    ref = rdec.LocalCondRNVPDecoder(L, f, G)
    mine = ours.LocalCondRNVPDecoder(L, f, G)
    load_into(ref, mine, 800)
    ref.eval()
    p, g = synth_inputs(B, N, G, 801)
    pt, gt = T(p).requires_grad_(True), T(g).requires_grad_(True)
    ps, mus, lvs = ref(pt, gt, mode='inverse')
    loss = 0.5 * (sum(lvs) + ps[0] ** 2).sum() / B
    loss.backward()
    out = {'dims': np.array([L, f, G, B, N]), 'p': p, 'g': g, 'loss': npy(loss), 'dp': npy(pt.grad), 'dg': npy(gt.grad)}
    named = dict(ref.named_parameters())
    for key in ('flows.0.nvp1.T_mu_0.mu_sd0.weight', 'flows.1.nvp2.T_logvar_0.logvar_sd1.weight',
                'flows.0.nvp3.T_mu_1.mu_sd2.weight', 'flows.1.nvp1.T_logvar_1.logvar_sd2.bias',
                'flows.0.nvp2.T_logvar_0_cond_w.logvar_sd1_film_w1.weight',
                'flows.1.nvp3.T_mu_0_cond_b.mu_sd1_film_b0.weight',
                'flows.0.nvp1.T_mu_0.mu_sd0_bn.weight', 'flows.0.nvp1.T_mu_0.mu_sd0_bn.bias'):
        out['grad::' + key] = npy(named[key].grad)
    save('g8_gradients', **out)


def g9_train_gradients():
    """Train-mode (batch-statistic BatchNorm) gradients of the genuine reference: loss.backward() as training.py:54."""
    L, f, G, B, N = 1, 8, 16, 3, 40
    ref = rdec.LocalCondRNVPDecoder(L, f, G)
    mine = ours.LocalCondRNVPDecoder(L, f, G)
    load_into(ref, mine, 900)
    ref.train()
    p, g = synth_inputs(B, N, G, 901)
    pt, gt = T(p).requires_grad_(True), T(g).requires_grad_(True)
    ps, mus, lvs = ref(pt, gt, mode='inverse')
    loss = 0.5 * (sum(lvs) + ps[0] ** 2).sum() / B
    loss.backward()
    out = {'dims': np.array([L, f, G, B, N]), 'p': p, 'g': g, 'loss': npy(loss), 'dp': npy(pt.grad), 'dg': npy(gt.grad),
           'z': npy(ps[0]), 'logdet': npy(sum(lvs))}
    for k, v in ref.named_parameters():
        out['grad::' + k] = npy(v.grad)
    save('g9_train_gradients', **out)


def g18_train_gradients_at_config_depth():
    """Train-mode gradients of the genuine reference at the airplane config's full depth (11 Triples = 33 couplings, f = 37,
    G = 128): every coupling's batch statistics depend on all couplings before it, so this is where errors of the backward
    pipeline would compound.  fp32 as the reference runs and fp64 for the noise floor.  Stored: outputs, loss, input gradients,
    the L2 norm of every parameter gradient, and the full gradients of three couplings (first, middle, last)."""
    L, f, G, B, N = 11, 37, 128, 3, 96
    p, g = synth_inputs(B, N, G, 1801)
    keep = ('flows.0.nvp1.', 'flows.5.nvp2.', 'flows.10.nvp3.')
    out = {'dims': np.array([L, f, G, B, N]), 'p': p, 'g': g}
    for dt, suffix in ((torch.float32, ''), (torch.float64, '_f64')):
        ref, mine = rdec.LocalCondRNVPDecoder(L, f, G), ours.LocalCondRNVPDecoder(L, f, G)
        load_into(ref, mine, 1800)
        ref = ref.to(dt).train()
        pt, gt = T(p).to(dt).requires_grad_(True), T(g).to(dt).requires_grad_(True)
        ps, mus, lvs = ref(pt, gt, mode='inverse')
        loss = 0.5 * (sum(lvs) + ps[0] ** 2).sum() / B
        loss.backward()
        out['loss' + suffix] = npy(loss).astype(np.float64)
        out['z' + suffix], out['logdet' + suffix] = npy(ps[0]).astype(np.float32), npy(sum(lvs)).astype(np.float32)
        out['dp' + suffix], out['dg' + suffix] = npy(pt.grad).astype(np.float32), npy(gt.grad).astype(np.float32)
        out['gnorm' + suffix] = np.array([float(v.grad.double().norm()) for _, v in ref.named_parameters()])
        for k, v in ref.named_parameters():
            if k.startswith(keep):
                out['grad' + suffix + '::' + k] = npy(v.grad).astype(np.float32)
    save('g18_train_depth_11x37x128', **out)


def g10_optimizer():
    """Three steps of the reference's Adam (both amsgrad settings, with weight decay and the LRUpdater schedule)."""
    from lib.networks.optimizers import Adam as RefAdam, LRUpdater
    rng = np.random.default_rng(1000)
    shapes = [(7,), (5, 3), (1, 4, 9), (130,)]
    out = {}
    for ams in (0, 1):
        ps = [torch.nn.Parameter(T(rng.normal(size=s).astype(np.float32))) for s in shapes]
        for i, q in enumerate(ps):
            out[f'p0_{ams}_{i}'] = npy(q).copy()
        opt = RefAdam(ps, lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-3, amsgrad=bool(ams))
        sched = LRUpdater(10, cycle_length=4, min_lr=1e-3, max_lr=1e-2, beta1=0.9, min_beta2=0.9, max_beta2=0.99)
        for step in range(3):
            sched(opt, 0, step)
            for i, q in enumerate(ps):
                gr = rng.normal(size=shapes[i]).astype(np.float32) * (10.0 if step == 1 else 1.0)
                out[f'g_{ams}_{step}_{i}'] = gr
                q.grad = T(gr.copy())
            opt.step()
            for i, q in enumerate(ps):
                out[f'p_{ams}_{step}_{i}'] = npy(q).copy()
        for i, q in enumerate(ps):
            out[f'm_{ams}_{i}'] = npy(opt.state[q]['exp_avg'])
            out[f'v_{ams}_{i}'] = npy(opt.state[q]['max_exp_avg_sq' if ams else 'exp_avg_sq'])
    save('g10_optimizer', **out)


def g11_encoder():
    """PointNet cloud encoder (+ max-pool, models.py:127-128) and the per-shape heads, eval and train BatchNorm."""
    from lib.networks import encoders as renc
    from go_with_the_flows_amd import encoders as oenc
    out = {}
    for tag, n_features, B, N, seed in (('big', [128, 256, 512], 3, 70, 1100), ('small', [128, 64, 128], 2, 300, 1110)):
        ref, mine = renc.PointNetCloudEncoder(3, 64, n_features), oenc.PointNetCloudEncoder(3, 64, n_features)
        x, _ = synth_inputs(B, N, 4, seed + 1)
        out[f'{tag}_x'] = x
        out[f'{tag}_dims'] = np.array([64] + n_features + [B, N])
        for training in (False, True):
            load_into(ref, mine, seed)
            ref.train(training)
            with torch.no_grad():
                feat = ref(T(x))
            t = 'train' if training else 'eval'
            out[f'{tag}_{t}_pooled'] = npy(torch.max(feat, dim=2)[0])
            out[f'{tag}_{t}_feat_head'] = npy(feat[:, :, :6])
            if training:
                sd = ref.state_dict()
                out[f'{tag}_rm_last'] = npy(sd[f'features.sd{len(n_features) - 1}_bn.running_mean'])
                out[f'{tag}_rv_last'] = npy(sd[f'features.sd{len(n_features) - 1}_bn.running_var'])
    # per-shape heads
    rng = np.random.default_rng(1120)
    h = rng.standard_normal((5, 48)).astype(np.float32)
    out['head_x'] = h
    for tag, cls_r, cls_o, kw in (('post', renc.FeatureEncoder, oenc.FeatureEncoder, dict(deterministic=False)),
                                  ('det', renc.FeatureEncoder, oenc.FeatureEncoder, dict(deterministic=True)),
                                  ('wts', renc.WeightsEncoder, oenc.WeightsEncoder, dict(deterministic=True))):
        ref, mine = cls_r(2, 48, 10, **kw), cls_o(2, 48, 10, **kw)
        for training in (False, True):
            load_into(ref, mine, 1130)
            ref.train(training)
            with torch.no_grad():
                y = ref(T(h))
            t = 'train' if training else 'eval'
            if isinstance(y, tuple):
                out[f'head_{tag}_{t}_mu'], out[f'head_{tag}_{t}_lv'] = npy(y[0]), npy(y[1])
            else:
                out[f'head_{tag}_{t}'] = npy(y)
    save('g11_encoder', **out)


def g12_prior():
    """Global prior flow on the latent (GlobalRNVPDecoder) and the Gaussian latent losses, as models.py:137-151 calls them."""
    from go_with_the_flows_amd import prior as oprior
    out = {}
    n_flows, F_, G, B = 3, 24, 16, 6
    ref, mine = rdec.GlobalRNVPDecoder(n_flows, F_, G), oprior.GlobalRNVPDecoder(n_flows, F_, G)
    out['dims'] = np.array([n_flows, F_, G, B])
    rng = np.random.default_rng(1200)
    g = rng.standard_normal((B, G)).astype(np.float32)
    mu0 = (0.1 * rng.standard_normal((1, G))).astype(np.float32)
    lv0 = (0.3 * rng.standard_normal((1, G))).astype(np.float32)
    post_lv = (0.5 * rng.standard_normal((B, G))).astype(np.float32)
    out.update(g=g, mu0=mu0, lv0=lv0, post_lv=post_lv)
    for training in (False, True):
        for mode in ('direct', 'inverse'):
            load_into(ref, mine, 1210)
            ref.train(training)
            with torch.no_grad():
                gs, mus, lvs = ref(T(g), mode=mode)
                t = f'{"train" if training else "eval"}_{mode}'
                out[f'gs_{t}'] = np.stack([npy(x) for x in gs])
                out[f'mus_{t}'] = np.stack([npy(x) for x in mus])
                out[f'lvs_{t}'] = np.stack([npy(x) for x in lvs])
                if mode == 'inverse':     # models.py:137-151 + losses.py: the training-mode latent loss terms
                    samples = gs + [T(g)]
                    all_mus = [T(mu0).expand(B, G)] + mus
                    all_lvs = [T(lv0).expand(B, G)] + lvs
                    out[f'gnll_{t}'] = npy(rloss.GaussianFlowNLL()(samples, all_mus, all_lvs))
                    out[f'gent_{t}'] = npy(rloss.GaussianEntropy()(T(post_lv)))
            if training:
                sd = ref.state_dict()
                out[f'rm_{t}'] = npy(sd['flows.1.nvp2.T_mu_0.mu_mlp0_bn.running_mean'])
                out[f'rv_{t}'] = npy(sd['flows.1.nvp2.T_mu_0.mu_mlp0_bn.running_var'])
    save('g12_prior', **out)


MODEL_CFG = dict(train_mode='p_rnvp_mc_g_rnvp_vae', util_mode='training', deterministic=False,
                 pc_enc_init_n_channels=3, pc_enc_init_n_features=64, pc_enc_n_features=[128, 64, 128],
                 g_latent_space_size=16, g_prior_n_flows=2, g_prior_n_features=16, g_posterior_n_layers=1,
                 p_latent_space_size=3, p_prior_n_layers=1, p_decoder_n_flows=2, p_decoder_n_features=8,
                 p_decoder_base_type='free', p_decoder_base_var=-3.9551, n_components=3,
                 params_reduce_mode='none', weights_type='learned_weights',
                 pnll_weight=1.0, gnll_weight=0.7, gent_weight=0.3)


def g13_full_model():
    """The whole Flow_Mixture_Model + Flow_Mixture_Loss with EVERY parameter seeded (state_dict contract of the full
    model), the reparameterisation noise recorded: training mode (eval and train BatchNorm) and labelled generation."""
    from go_with_the_flows_amd import models as omodels
    out = {}
    B, N, G, K = 4, 48, MODEL_CFG['g_latent_space_size'], MODEL_CFG['n_components']
    rng = np.random.default_rng(1300)
    gcloud, _ = synth_inputs(B, N, G, 1301)
    pcloud, _ = synth_inputs(B, N, G, 1302)
    noise_g = rng.standard_normal((B, G)).astype(np.float32)
    out.update(gcloud=gcloud, pcloud=pcloud, noise_g=noise_g, dims=np.array([B, N, G, K]))
    keys = None
    for base_type in ('free', 'freevar'):
        cfg = dict(MODEL_CFG, p_decoder_base_type=base_type)
        ref, mine = Flow_Mixture_Model(**cfg), omodels.Flow_Mixture_Model(**cfg)
        ref.reparameterize = lambda mu, logvar: T(noise_g) * torch.exp(0.5 * logvar) + mu
        for training in (False, True):
            load_into(ref, mine, 1310)
            ref.train(training)
            with torch.no_grad():
                enc, dec, logits = ref(T(gcloud), T(pcloud), None, None, False, False)
                loss, pnll, gnll, gent = rloss.Flow_Mixture_Loss(**cfg)(enc, dec, logits)
            t = f'{base_type}_{"train" if training else "eval"}'
            out[f'terms_{t}'] = np.array([float(loss), float(pnll), float(gnll), float(gent)])
            out[f'logits_{t}'] = npy(logits)
            out[f'g_sample_{t}'] = npy(enc['g_posterior_samples'])
            out[f'g_base_{t}'] = npy(enc['g_prior_samples'][0])
            out[f'z_{t}'] = np.stack([npy(o['p_prior_samples'][0]) for o in dec])
            out[f'lv0_{t}'] = np.stack([npy(o['p_prior_logvars'][0][:, :, 0]) for o in dec])
            out[f'n_lists_{t}'] = np.array([len(enc['g_prior_samples']), len(enc['g_prior_mus']), len(dec[0]['p_prior_samples'])])
            if training:
                sd = ref.state_dict()
                out[f'rm_pprior_{t}'] = npy(sd['p_prior.features.mlp0_bn.running_mean'])
        if base_type == 'free':
            keys = [[k, list(v.shape), str(v.dtype).replace('torch.', '')] for k, v in ref.state_dict().items()]
    # labelled generation of ONE shape (flow_mixture.py:146-177): numpy draw seeded, base noise recorded
    cfg = dict(MODEL_CFG, util_mode='generating')
    ref, mine = Flow_Mixture_Model(**cfg), omodels.Flow_Mixture_Model(**cfg)
    load_into(ref, mine, 1310)
    ref.eval()
    Ns = 40
    noise_p = rng.standard_normal((1, 3, Ns)).astype(np.float32)
    noise_g1 = rng.standard_normal((1, G)).astype(np.float32)

    def rep(mu, logvar):
        if mu.dim() == 2:
            return T(noise_g1) * torch.exp(0.5 * logvar) + mu
        return T(noise_p[:, :, :mu.shape[2]]) * torch.exp(0.5 * logvar) + mu
    ref.reparameterize = rep
    np.random.seed(1320)
    with torch.no_grad():
        enc, samples, labels, logits = ref(T(gcloud[:1, :, :Ns]), T(pcloud[:1, :, :Ns]), None, Ns, True, False)
    out.update(gen_noise_p=noise_p, gen_noise_g=noise_g1, gen_samples=npy(samples), gen_labels=npy(labels),
               gen_logits=npy(logits), gen_g=npy(enc['g_prior_samples'][-1]))
    save('g13_full_model', **out)
    with open(os.path.join(HERE, 'contract_model.json'), 'w') as fh:
        json.dump({'cfg': MODEL_CFG, 'state_dict': keys}, fh, indent=0)
    print('contract_model.json', len(keys), 'entries')


def g19_autoencoding():
    """Labelled reconstruction of ONE shape in 'autoencoding' mode (what evaluate_ae.py runs for the autoencoding configs:
    posterior mean as the shape code, models.py:111-133; per-point component draw + direct decoding, flow_mixture.py:146-177),
    numpy draw seeded, base noise recorded."""
    from go_with_the_flows_amd import models as omodels
    cfg = dict(MODEL_CFG, util_mode='autoencoding')
    ref, mine = Flow_Mixture_Model(**cfg), omodels.Flow_Mixture_Model(**cfg)
    load_into(ref, mine, 1310)
    ref.eval()
    G, Ns = MODEL_CFG['g_latent_space_size'], 40
    gcloud, _ = synth_inputs(4, 48, G, 1301)
    pcloud, _ = synth_inputs(4, 48, G, 1302)
    rng = np.random.default_rng(1900)
    noise_p = rng.standard_normal((1, 3, Ns)).astype(np.float32)
    ref.reparameterize = lambda mu, logvar: T(noise_p[:, :, :mu.shape[2]]) * torch.exp(0.5 * logvar) + mu
    np.random.seed(1920)
    with torch.no_grad():
        enc, samples, labels, logits = ref(T(gcloud[1:2, :, :Ns]), T(pcloud[1:2, :, :Ns]), None, Ns, True, False)
    save('g19_autoencoding', gcloud=gcloud[1:2, :, :Ns], pcloud=pcloud[1:2, :, :Ns], noise_p=noise_p, samples=npy(samples),
         labels=npy(labels), logits=npy(logits), g_code=npy(enc['g_posterior_mus']),
         n_lists=np.array([len(enc['g_prior_samples']), len(enc['g_prior_mus'])]))


def _pure_functions(relpath, names):
    """Execute the named top-level functions of a reference file from its own source, skipping the imports of the
    un-built CUDA extension (lib/metrics/StructuralLosses: nvcc-only) that keep the whole module from importing here.
    Nothing is stubbed: a function that reaches the extension simply is not exercised."""
    import ast
    import types
    path = os.path.join(REF, relpath)
    tree = ast.parse(open(path).read(), path)
    body = []
    for node in tree.body:
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            text = ast.unparse(node)
            if 'StructuralLosses' in text or 'sklearn' in text:
                continue
            body.append(node)
        elif isinstance(node, ast.FunctionDef) and node.name in names:
            body.append(node)
    mod = types.ModuleType('ref_' + os.path.basename(relpath)[:-3])
    exec(compile(ast.Module(body=body, type_ignores=[]), path, 'exec'), mod.__dict__)
    return mod


def g14_evaluation_metrics():
    """Host-side evaluation metrics of the reference (lib/metrics/evaluation_metrics.py, lib/networks/utils.py) on the
    paths that do not enter its CUDA extension: pure-torch Chamfer, F1, MMD / coverage / 1-NN bookkeeping, occupancy JSD."""
    em = _pure_functions('lib/metrics/evaluation_metrics.py',
                         {'distChamfer', 'EMD_CD_F1', '_pairwise_EMD_CD_F1_SCORE', 'knn', 'lgan_mmd_cov', 'compute_all_metrics',
                          'distChamferCUDA', 'emd_approx'})
    ut = _pure_functions('lib/networks/utils.py', {'get_voxel_occ_dist', 'JSD'})
    rng = np.random.default_rng(1400)
    a = (0.3 * rng.standard_normal((3, 40, 3))).astype(np.float32)
    b = (0.3 * rng.standard_normal((3, 40, 3))).astype(np.float32)
    out = dict(a=a, b=b)
    first, second = em.distChamfer(T(a), T(b))
    out.update(chamfer_first=npy(first), chamfer_second=npy(second))
    smp = (rng.random((6, 48, 3)) - 0.5).astype(np.float32)
    ref = (rng.random((6, 48, 3)) - 0.5).astype(np.float32)
    out.update(smp=smp, ref=ref)
    r = em.EMD_CD_F1(T(smp), T(ref), 4, accelerated_cd=False, reduced=False, cd_option=True, one_part_of_cd=True,
                     f1_option=True, f1_threshold=0.01)
    out.update(pair_CD=npy(r['CD']), pair_CDL=npy(r['CDL']), pair_CDR=npy(r['CDR']), pair_F1=npy(r['F1']))
    r = em.EMD_CD_F1(T(smp), T(ref), 4, accelerated_cd=False, reduced=True, cd_option=True, f1_option=True, f1_threshold=0.01)
    out.update(pair_CD_mean=npy(r['CD']), pair_F1_mean=npy(r['F1']))
    Mxx, Mxy, Myy = (torch.from_numpy(rng.random((5, 5)).astype(np.float32)) for _ in range(3))
    Mxx, Myy = Mxx + Mxx.t(), Myy + Myy.t()
    out.update(Mxx=npy(Mxx), Mxy=npy(Mxy), Myy=npy(Myy))
    for k, sq in ((1, False), (3, True)):
        res = em.knn(Mxx, Mxy, Myy, k, sqrt=sq)
        out.update({f'knn{k}_{key}': npy(v) for key, v in res.items()})
    D = torch.from_numpy(rng.random((6, 4)).astype(np.float32))
    out['lgan_in'] = npy(D)
    for mode in ('min', 'max'):
        res = em.lgan_mmd_cov(D, mode)
        out.update({f'lgan_{mode}_{key}': npy(v) for key, v in res.items()})
    res = em.compute_all_metrics(T(smp), T(ref[:5]), 4, accelerated_cd=False, f1_threshold=0.01, cd_option=True,
                                 one_part_of_cd=True, f1_option=True, emd_option=False)
    names = sorted(res)
    out['all_names'] = np.array(names)
    for key in names:
        out['all_' + key] = npy(res[key]) if torch.is_tensor(res[key]) else np.asarray(res[key])
    c1 = (rng.random((9, 64, 3)) - 0.5).astype(np.float32) * 0.98
    c2 = (0.25 * rng.standard_normal((7, 64, 3))).astype(np.float32).clip(-0.6, 0.6)   # some points outside the cube
    out.update(c1=c1, c2=c2, occ1=ut.get_voxel_occ_dist(c1, warning=False), occ2=ut.get_voxel_occ_dist(c2, warning=False),
               jsd=np.float64(ut.JSD(c1, c2, warning=False)))
    save('g14_evaluation', **out)


def contract():
    """Reference state_dict keys/shapes for a small decoder, as JSON (checkpoint contract)."""
    ref = rdec.LocalCondRNVPDecoder(2, 8, 16)
    spec = [[k, list(v.shape), str(v.dtype).replace('torch.', '')] for k, v in ref.state_dict().items()]
    with open(os.path.join(HERE, 'contract.json'), 'w') as fh:
        json.dump({'ctor': [2, 8, 16], 'state_dict': spec}, fh, indent=0)
    print('contract.json', len(spec), 'entries')


def g17_encoder_train():
    """The reference PointNet encoder + max-pool under model.train(): pooled code, every BatchNorm's running statistics
    after one step and the gradient of a weighted sum of the pooled code w.r.t. all 12 parameter tensors (fp32 as the
    reference runs, and the same in fp64 for the noise floor).  N a multiple of 4: what the HIP train pipeline covers."""
    from lib.networks import encoders as renc
    from go_with_the_flows_amd import encoders as oenc
    out = {}
    for tag, B, N, seed in (('a', 3, 72, 1700), ('b', 2, 260, 1710)):
        x, _ = synth_inputs(B, N, 4, seed + 1)
        wgt = np.random.default_rng(seed + 2).standard_normal((B, 512)).astype(np.float32)
        out[f'{tag}_x'], out[f'{tag}_wgt'] = x, wgt
        for dt, suffix in ((torch.float32, ''), (torch.float64, '_f64')):
            ref, mine = renc.PointNetCloudEncoder(3, 64, [128, 256, 512]), oenc.PointNetCloudEncoder(3, 64, [128, 256, 512])
            load_into(ref, mine, seed)
            ref = ref.to(dt).train()
            pooled = torch.max(ref(T(x).to(dt)), dim=2)[0]
            (pooled * T(wgt).to(dt)).sum().backward()
            out[f'{tag}_pooled{suffix}'] = npy(pooled).astype(np.float64 if suffix else np.float32)
            for name, prm in ref.named_parameters():
                out[f'{tag}_grad{suffix}.{name}'] = npy(prm.grad).astype(np.float32)     # fp64 run rounded once
            if not suffix:
                for name, buf in ref.named_buffers():
                    out[f'{tag}_buf.{name}'] = npy(buf)
    save('g17_encoder_train', **out)


def g20_heads():
    """The per-shape heads at the sizes the shipped configs give them (models.py:51-59, flow_mixture.py:28-32): g_posterior =
    FeatureEncoder(1, 512, 128), p_prior = FeatureEncoder(1, 128, 3), mixture_weights_encoder = WeightsEncoder(3, 128, 4,
    deterministic) -- the genuine reference modules (encoders.py:31-91) in train mode (batch statistics, running statistics after
    the step) and eval mode: outputs, and the gradient of a weighted sum of the outputs w.r.t. the input and every parameter,
    fp32 as the reference runs and fp64 for the noise floor."""
    from lib.networks import encoders as renc
    from go_with_the_flows_amd import encoders as oenc
    out = {}
    cases = (('post', renc.FeatureEncoder, oenc.FeatureEncoder, (1, 512, 128), dict(deterministic=False), 7, 2000),
             ('pprior', renc.FeatureEncoder, oenc.FeatureEncoder, (1, 128, 3), dict(deterministic=False), 64, 2010),
             ('wts', renc.WeightsEncoder, oenc.WeightsEncoder, (3, 128, 4), dict(deterministic=True), 33, 2020))
    for tag, cls_r, cls_o, ctor, kw, B, seed in cases:
        rng = np.random.default_rng(seed + 1)
        x = rng.standard_normal((B, ctor[1])).astype(np.float32)
        n_out = 1 if kw['deterministic'] else 2
        wgt = rng.standard_normal((n_out, B, ctor[2])).astype(np.float32)
        out[f'{tag}_x'], out[f'{tag}_wgt'], out[f'{tag}_ctor'] = x, wgt, np.array(ctor)
        for training in (True, False):
            t = 'train' if training else 'eval'
            res = {}
            for dt, suffix in ((torch.float32, 'f32'), (torch.float64, 'f64')):
                ref, mine = cls_r(*ctor, **kw), cls_o(*ctor, **kw)
                load_into(ref, mine, seed)
                ref = ref.to(dt).train(training)
                xin = T(x).to(dt).requires_grad_(True)
                y = ref(xin)
                ys = y if isinstance(y, tuple) else (y,)
                sum((o * T(wgt[i]).to(dt)).sum() for i, o in enumerate(ys)).backward()
                res[suffix] = dict({f'out{i}': npy(o) for i, o in enumerate(ys)}, gx=npy(xin.grad),
                                   **{f'grad.{name}': npy(prm.grad) for name, prm in ref.named_parameters()})
                if training and suffix == 'f32':
                    for name, buf in ref.named_buffers():
                        out[f'{tag}_buf.{name}'] = npy(buf)
            # stored: the fp64 run (rounded to fp32 once; large weight gradients every 8th row) and, per tensor, the distance of
            # the reference's own fp32 run from it relative to the tensor's largest entry -- the noise floor a test may allow
            for key, v64 in res['f64'].items():
                v32 = res['f32'][key]
                out[f'{tag}_{t}_noise.{key}'] = np.float64(np.abs(v32 - v64).max() / max(np.abs(v64).max(), 1e-30))
                out[f'{tag}_{t}_{key}'] = (v64[::8] if v64.size > 65536 else v64).astype(np.float32)
    save('g20_heads', **out)


CASES = {
    'contract': contract,
    'g1': g1_single_couplings,
    'g2': g2_triples,
    'g3': lambda: decoder_case('g3_decoder_4x64x128', 4, 64, 128, 4, 512, 400, full_lists=False, train_too=True),
    'g3s': lambda: decoder_case('g3s_decoder_lists', 2, 8, 16, 2, 24, 410, full_lists=True, train_too=True),
    'g4_37': lambda: decoder_case('g4_width37', 2, 37, 128, 2, 64, 420, full_lists=False, train_too=False),
    'g4_33': lambda: decoder_case('g4_width33', 2, 33, 512, 2, 64, 430, full_lists=False, train_too=False),
    'g4_19': lambda: decoder_case('g4_width19', 2, 19, 128, 2, 64, 440, full_lists=False, train_too=False),
    'g5': g5_losses,
    'g6': g6_keep_drift,
    'g7': g7_model_training_forward,
    'g8': g8_gradients,
    'g9': g9_train_gradients,
    'g18': g18_train_gradients_at_config_depth,
    'g10': g10_optimizer,
    'g11': g11_encoder,
    'g17': g17_encoder_train,
    'g20': g20_heads,
    'g12': g12_prior,
    'g13': g13_full_model,
    'g19': g19_autoencoding,
    'g14': g14_evaluation_metrics,
    # the decoders of BASELINE.json's configs at their FULL depth (33 / 33 / 18 couplings), genuine reference, fp32 + fp64
    'g15_airplane': lambda: decoder_case('g15_depth_11x37x128', 11, 37, 128, 2, 256, 1500, full_lists=False, train_too=False),
    'g15_ae': lambda: decoder_case('g15_depth_11x33x512', 11, 33, 512, 2, 128, 1510, full_lists=False, train_too=False),
    'g15_k16': lambda: decoder_case('g15_depth_6x19x128', 6, 19, 128, 2, 256, 1520, full_lists=False, train_too=False),
    # widths beyond 64 (the reference accepts any f_n_features, flows.py:11-16): 96 = train / backward limit, 128 = forward limit, 80
    'g16_w80': lambda: decoder_case('g16_width80', 1, 80, 32, 3, 96, 1600, full_lists=False, train_too=True),
    'g16_w96': lambda: decoder_case('g16_width96', 2, 96, 64, 3, 80, 1610, full_lists=False, train_too=True),
    'g16_w128': lambda: decoder_case('g16_width128', 1, 128, 64, 2, 80, 1620, full_lists=False, train_too=False),
    'g16_w100': lambda: decoder_case('g16_width100', 1, 100, 32, 2, 70, 1630, full_lists=False, train_too=False),
}

if __name__ == '__main__':
    # no arguments: regenerate everything; otherwise only the named cases (see CASES)
    for name in (sys.argv[1:] or list(CASES)):
        CASES[name]()
