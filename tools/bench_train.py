#!/usr/bin/env python3
"""End-to-end training step of the airplane generative config (config_generative_modeling_airplane.yaml: K=4 components,
decoders resolve to 11 Triples x f=37, G=128, B=64 x N=2048) on one MI355X: encoder + posterior + prior flow + 4 decoders
(batch-statistic BatchNorm) + mixture NLL + backward + fused Adam.  GPU box only.

    python tools/bench_train.py [--batch 64] [--steps 5] [--graph]

Data parallel: started once per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment (torch.distributed.run, or
bench.py's own per-rank children) it joins an RCCL process group, converts the model to SyncBatchNorm (train_ae.py:152) and runs
the sharded step -- statistic all-reduces, row all-gathers and the overlapped gradient exchange all inside the one hipGraph;
--batch is the PER-RANK batch.  GWTF_FORCE_SHARDED=1 takes that path on a 1-rank group (one-GPU box).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from go_with_the_flows_amd import models, optim
from go_with_the_flows_amd.synth import load_synth_, synth_inputs

ap = argparse.ArgumentParser()
ap.add_argument('--config', default='airplane', choices=['airplane', 'ae'],
                help='airplane: config_generative_modeling_airplane.yaml (G=128, base free, learned weights); ae: config_autoencoding.yaml '
                     '(G=512, base freevar: decoders resolve to 11 Triples x f=33) -- BASELINE configs[1] / configs[2]')
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--points', type=int, default=2048)
ap.add_argument('--steps', type=int, default=5)
ap.add_argument('--graph', action='store_true', help='capture forward + backward in one hipGraph')
ap.add_argument('--api', default='fused', choices=['fused', 'list', 'both', 'swap', 'swap_loop'],
                help="fused: forward_fused + Flow_Mixture_Loss.fused; list: the reference's own call, model(g, p) -> lists -> loss; "
                     "both: fused, then a second graph of the list call in the same process; swap: the literal import swap of "
                     "INTEGRATION.md section 1 -- the K decoders called ONE AT A TIME (flow_mixture.py:163-166) and the loss on their "
                     "lists; swap_loop: the same with the loss as the reference's B x K Python loop (losses.py:109-131, restated "
                     "below).  GWTF_NO_SIBLING_BATCH=1 turns the decoders' sibling batching off (what round 4 shipped)")
ap.add_argument('--torch-profile', type=int, default=0, metavar='ROWS',
                help='after the warm-up: torch.profiler over 3 eager steps, the ROWS operators with the most device time (with shapes), exit')
ap.add_argument('--lib', default=None, help='A/B: load this build of libgwtf_hip.so instead of the in-tree one')
ap.add_argument('--parts', default='epd', help='debug: which parts run (e=encoder, p=prior flow, d=decoders)')
ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'], help='process-group backend of a data-parallel run (nccl = RCCL)')
ap.add_argument('--share-device', action='store_true', help='rehearsal: every rank uses cuda:0 (needs --backend gloo; no graph)')
a = ap.parse_args()
if a.lib:
    from go_with_the_flows_amd import _lib
    _lib.LIB_PATH = a.lib
WORLD, RANK = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
FORCED = os.environ.get('GWTF_FORCE_SHARDED') == '1'
torch.cuda.set_device(0 if a.share_device else int(os.environ.get('LOCAL_RANK', '0')))
if WORLD > 1 or FORCED:
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(29900 + os.getpid() % 90))
    if a.backend == 'nccl':
        dist.init_process_group('nccl', rank=RANK, world_size=WORLD, device_id=torch.device('cuda', torch.cuda.current_device()))
    else:
        dist.init_process_group('gloo', rank=RANK, world_size=WORLD)
say = print if RANK == 0 else (lambda *x, **k: None)

CFG = dict(train_mode='p_rnvp_mc_g_rnvp_vae', util_mode='training', deterministic=False,
           pc_enc_init_n_channels=3, pc_enc_init_n_features=64, pc_enc_n_features=[128, 256, 512],
           g_latent_space_size=128, g_prior_n_flows=7, g_prior_n_features=128, g_posterior_n_layers=1,
           p_latent_space_size=3, p_prior_n_layers=1, p_decoder_n_flows=21, p_decoder_n_features=64,
           p_decoder_base_type='free', p_decoder_base_var=-3.9551, n_components=4,
           params_reduce_mode='depth_and_feature', weights_type='learned_weights',
           pnll_weight=1.0, gnll_weight=1.0, gent_weight=1.0)
if a.config == 'ae':        # config_autoencoding.yaml:19-43
    CFG.update(g_latent_space_size=512, p_decoder_base_type='freevar', p_decoder_base_var=-3.596)   # weights_type: the CLI flag of scripts/train_*.sh
torch.manual_seed(0)
model = models.Flow_Mixture_Model(**CFG).cuda().train()
if WORLD > 1 or FORCED:
    model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)          # train_ae.py:152
crit = models.Flow_Mixture_Loss(**CFG)
opt = optim.Adam(model.parameters(), lr=2.56e-4, betas=(0.9, 0.999), weight_decay=1e-6 if a.config == 'ae' else 1e-5, amsgrad=True)
n_params = sum(p.numel() for p in model.parameters())
say(f'decoders: {len(model.pc_decoder)} x ({model.pc_decoder[0].n_flows} Triples, f={model.pc_decoder[0].f_n_features}); '
      f'{n_params / 1e6:.2f} M parameters')
g_in = torch.from_numpy(synth_inputs(a.batch, a.points, 4, 1 + 10 * RANK)[0]).cuda()          # every rank its own shapes
p_in = torch.from_numpy(synth_inputs(a.batch, a.points, 4, 2 + 10 * RANK)[0]).cuda()
from go_with_the_flows_amd import autograd as gwa
from go_with_the_flows_amd.dist import OverlappedGradients, graph_capture, sharded
reducer = OverlappedGradients(model) if sharded() else None


g_fix = torch.randn(a.batch, CFG['g_latent_space_size'], device='cuda')


API = [a.api if a.api in ('list', 'swap', 'swap_loop') else 'fused']
if a.api in ('swap', 'swap_loop'):
    model.literal_k_loop = True


def loop_mixture_nll(output_decoder, logits):
    """The point NLL as the reference evaluates it (losses.py:101-137), restated: a Python loop over the B shapes and the K
    components that re-adds every component's whole list of log-variances (`sum(list)` of 3 n_flows + 1 full (B,3,N) tensors) for
    each of its B x K iterations.  Timing harness for `--api swap_loop`; its value equals Flow_Mixture_Loss's (asserted below)."""
    log_w = (logits - torch.logsumexp(logits, dim=-1, keepdim=True)).unsqueeze(1)              # (B, 1, K)
    K, B = len(output_decoder), output_decoder[0]['p_prior_mus'][0].shape[0]
    per_shape = []
    for i in range(B):
        cols = []
        for j in range(K):
            o = output_decoder[j]
            mu0, lv0 = o['p_prior_mus'][0][i], o['p_prior_logvars'][0][i]
            logdet = sum(o['p_prior_logvars'])[i]
            z = o['p_prior_samples'][0][i]
            inner = -torch.sum(logdet + (z - mu0) ** 2 / torch.exp(lv0), dim=0, keepdim=True)
            cols.append(0.5 * (inner - float(np.log(2.0 * np.pi)) * z.shape[0]))
        lp = torch.cat(cols, dim=0).t() + log_w[i]
        per_shape.append(-torch.logsumexp(lp, dim=-1).sum().unsqueeze(0))
    return torch.cat(per_shape).mean()


def fwd_bwd():
    opt.zero_grad(set_to_none=True)
    if a.parts == 'epd' and API[0] == 'swap_loop':
        output_prior, output_decoder, logits = model(g_in, p_in)
        loss, pnll, gnll, gent = crit._combine(loop_mixture_nll(output_decoder, logits), output_prior)
    elif a.parts == 'epd' and API[0] in ('list', 'swap'):
        output_prior, output_decoder, logits = model(g_in, p_in)             # training.py:43-47 of the reference
        loss, pnll, gnll, gent = crit(output_prior, output_decoder, logits)
    elif a.parts == 'epd':
        enc, dec = model.forward_fused(g_in, p_in)
        loss, pnll, gnll, gent = crit.fused(enc, dec)
    else:                                   # debug: isolate one part of the step
        loss = 0.0
        if 'e' in a.parts:
            loss = loss + model.g_posterior(model.pc_encoder.forward_max(g_in))[0].sum()
        if 'r' in a.parts:
            loss = loss + torch.randn_like(g_fix).sum() * model.g0_prior_mus.sum()
        if 'p' in a.parts:
            gs, mus, lvs = model.g_prior(g_fix, mode='inverse')
            loss = loss + gs[0].square().sum() + sum(lvs).sum()
        if 'd' in a.parts:
            z, ld = model.mixture_stack().forward_all(p_in, g_fix, mode='inverse')
            loss = loss + (z.square().sum() + ld.sum()) / a.batch
        if 'n' in a.parts:
            z, ld = model.mixture_stack().forward_all(p_in, g_fix, mode='inverse')
            mu0 = torch.zeros(4, a.batch, 3, device='cuda'); lv0 = torch.zeros(4, a.batch, 3, device='cuda')
            from go_with_the_flows_amd.mixture import flow_mixture_nll
            loss = loss + flow_mixture_nll(z, ld, mu0, lv0, model.get_weights(g_fix))[0]
    if reducer is not None:
        with reducer:              # the gradient exchange: one asynchronous all-reduce per decoder from inside the backward + a remainder
            loss.backward()
    else:
        loss.backward()
    return loss


def step():
    loss = fwd_bwd()
    opt.step()
    return loss


def timed(fn, n):
    torch.cuda.synchronize()
    if WORLD > 1:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    dt = torch.tensor([(time.perf_counter() - t0) / n * 1e3], dtype=torch.float64, device='cuda' if a.backend == 'nccl' else 'cpu')
    if WORLD > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)          # the slowest rank's time
    return float(dt)


for _ in range(2):
    l = step()
say('loss after warm-up', float(l.detach()))
del l                       # keep no reference to an autograd graph across iterations (hipGraph capture needs that)
if a.torch_profile:
    from torch.profiler import ProfilerActivity, profile
    step()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    print(prof.key_averages(group_by_input_shape=True).table(sort_by='self_cuda_time_total', row_limit=a.torch_profile,
                                                             max_name_column_width=60, max_shapes_column_width=90))
    sys.exit(0)
ms = timed(step, a.steps)
pts = a.batch * a.points * WORLD
say(f'ranks: {WORLD}  per-rank batch: {a.batch}  sharded path: {sharded()}')
_rows = a.batch * (WORLD if sharded() else 1)      # rows the per-shape modules see (all ranks' rows when data parallel)
_Gl = CFG['g_latent_space_size']
_heads_hip = all(m._hip_layers(torch.zeros(_rows, m.features[0].weight.shape[1] if m.n_layers else 1, device='cuda')) is not None
                 for m in (model.g_posterior, model.p_prior) if m.n_layers)
_prior_hip = model.g_prior._fused_ok(torch.zeros(a.batch, _Gl, device='cuda'), _rows)
_film = '+'.join(k.split('_')[-1] for k, v in gwa.PATHS.items() if v) or 'not run'      # the implementation(s) the steps above went through
say(f'per-shape modules: rows={_rows} film_heads={_film} heads={"hip" if _heads_hip else "library"} prior_flow={"hip" if _prior_hip else "library"}')
say(f'eager   : {ms:8.2f} ms/step  {pts / ms / 1e3:8.2f} Mpoints/s (each point through all {CFG["n_components"]} components)')
if a.graph:
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    gwa.COLLECTIVES['n'] = 0
    with graph_capture(graph):
        fwd_bwd()          # the optimiser stays outside: its bias corrections depend on the host-side step count
    say(f'statistic all-reduces captured in the graph: {gwa.COLLECTIVES["n"]}')

    def replay_step():
        graph.replay()
        opt.step()
    ms = timed(replay_step, a.steps)
    say(f'hipGraph: {ms:8.2f} ms/step  {pts / ms / 1e3:8.2f} Mpoints/s')
    if a.api == 'both':
        API[0] = 'list'
        with torch.cuda.stream(s):
            for _ in range(2):
                step()
        torch.cuda.current_stream().wait_stream(s)
        graph_list = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        with graph_capture(graph_list):
            fwd_bwd()

        def replay_list():
            graph_list.replay()
            opt.step()
        ms = timed(replay_list, a.steps)
        say(f'hipGraph list API: {ms:8.2f} ms/step  {pts / ms / 1e3:8.2f} Mpoints/s')
if WORLD > 1 or FORCED:
    dist.barrier()
    dist.destroy_process_group()
