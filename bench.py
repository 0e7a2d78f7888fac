#!/usr/bin/env python3
"""Headline benchmark: point-flow forward + log-det throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload airplane|m1|ae|svr|k16|k16_b1]

One "step" = one pass of the hot path over one batch of synthetic clouds that are already resident in
HBM: for every mixture component, the FiLM kernel + the fused coupling-stack kernel (inverse direction,
eval-mode BatchNorm), producing the base-space coordinates and the per-coordinate sum of logvars.
Weight packing is module preparation (cached while parameters are unchanged) and is outside the step,
like BatchNorm folding for inference.  Metric (BASELINE.json): Mpoints/s, one "point" = one 3-D point
pushed through one component's full stack.  N>1: one process per GPU, batch of shapes sharded, per-GPU batch
fixed (weak scaling), no data-path collective; started either by the driver's `python -m
torch.distributed.run ... bench.py --gpus N` or, when no launcher environment is present, by bench.py itself
(`python bench.py --gpus N` spawns the N ranks before touching the GPU, like the reference's mp.spawn in train_ae.py:183-193).

Prints ONE JSON line on rank 0: the BASELINE metric on configs[1] (`value`), `roofline` of the dominant kernel against
the unit that executes it (kernel duration from HIP events around the launch; `traffic` = HBM-side bytes per launch measured by two
`rocprofv3 --pmc` child passes of this script, N = 1 only: headline and `also.m1`, null elsewhere), `also` = the other BASELINE shapes
(m1, ae, svr, k16, k16_b1) timed by the same protocol in the same run plus the whole training step (`train_step`: children of this
process, one per rank; N = 1: plain, the reference's list API, the data-parallel code path on a 1-rank RCCL group, and configs[2]'s
16-shape per-rank shard), and `cpu_baseline`.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import go_with_the_flows_amd as gw                      # noqa: E402
from go_with_the_flows_amd import _lib                  # noqa: E402
from go_with_the_flows_amd.synth import load_synth_, synth_inputs  # noqa: E402
from go_with_the_flows_amd.dist import graph_capture      # noqa: E402

# Resolved dimensions of BASELINE.json's configs: K components, L triples, f, G, per-GPU B, N
WORKLOADS = {
    'airplane': dict(K=4, L=11, f=37, G=128, B=64, N=2048, mode='inverse',
                     name='configs[1] airplane generative (config_generative_modeling_airplane.yaml): '
                          'K=4 flows x 33 couplings, f=37, G=128, B=64 x N=2048, inverse + sum(logvars) + mixture NLL, eval BN'),
    'm1': dict(K=1, L=4, f=64, G=128, B=32, N=2048, mode='inverse',
               name='north-star shape: single flow, 12 couplings, f=64, G=128, B=32 x N=2048, inverse + sum(logvars)'),
    'ae': dict(K=4, L=11, f=33, G=512, B=16, N=2048, mode='inverse',
               name='configs[2] autoencoding per-GPU shard: K=4 x 33 couplings, f=33, G=512, B=16 x N=2048'),
    'svr': dict(K=4, L=11, f=33, G=512, B=16, N=2500, mode='inverse',
                name='configs[4] single-view reconstruction per-GPU shard (decoder side; the image encoder is out of scope): '
                     'K=4 x 33 couplings, f=33, G=512, B=16 x N=2500'),
    'k16': dict(K=16, L=6, f=19, G=128, B=32, N=2048, mode='direct',
                name='configs[3] K=16 mixture sampling, batched: each point visits one of 16 flows, 18 couplings, f=19'),
    'k16_b1': dict(K=16, L=6, f=19, G=128, B=1, N=2048, mode='direct',
                   name='configs[3] K=16 mixture sampling with the reference\'s own batch of ONE shape (flow_mixture.py:146): '
                        'latency of one 2048-point sample'),
}
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_* dense peak (= fp32 vector peak)
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense f16/bf16 MFMA peak; the contraction runs as 3 f16 products per fp32 product


def flops_per_point(L, f):
    """Algorithmic flop per point per component: C*(4f^2 + 12f), unpadded f."""
    return 3 * L * (4 * f * f + 12 * f)


def host_cores():
    """CPU cores this process may actually use: affinity mask capped by the cgroup quota (a GPU box gives
    each job a share of a much larger host; oversubscribing the torch thread pool is catastrophically slow)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.lower().startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(cfg, budget_s=15.0):
    """PyTorch-CPU port of the reference forward (oracle/torch_port.py) on a bounded sample of the workload."""
    from oracle import torch_port as tp
    torch.set_num_threads(host_cores())
    dec = gw.LocalCondRNVPDecoder(cfg['L'], cfg['f'], cfg['G'])
    st = load_synth_(dec, 2)
    tst = {k: torch.from_numpy(v) for k, v in st.items()}
    bs = min(cfg['B'], 8)
    p, g = synth_inputs(bs, cfg['N'], cfg['G'], 0)
    pt, gt = torch.from_numpy(p), torch.from_numpy(g)
    tp.decoder_fused(pt, gt, tst, cfg['L'], cfg['mode'])      # warm-up
    times, t_start = [], time.perf_counter()
    while len(times) < 3 or (time.perf_counter() - t_start < budget_s and len(times) < 50):
        t0 = time.perf_counter()
        tp.decoder_fused(pt, gt, tst, cfg['L'], cfg['mode'])
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    rate = lambda t: round(bs * cfg['N'] / t / 1e6, 4)
    return {'value': rate(med), 'unit': 'Mpoints/s', 'cores': torch.get_num_threads(),
            'kind': 'port', 'cpu_model': cpu_model(), 'min': rate(max(times)), 'max': rate(min(times)), 'reps': len(times),
            'sample': f'oracle/torch_port.py (PyTorch-CPU restatement of the reference forward, pinned to its golden '
                      f'vectors), one component, {bs} shapes x {cfg["N"]} points, median of {len(times)} reps '
                      f'(min / max = slowest / fastest rep)'}


def _run_bench_train(extra_args, env, timeout=200, graph=True, steps=20):
    """One run of tools/bench_train.py in a child process -> {label: ms per step} parsed from its report, or {'error': ...}."""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    try:
        r = subprocess.run([sys.executable, os.path.join(here, 'tools', 'bench_train.py'), '--steps', str(steps)] + (['--graph'] if graph else []) + extra_args,
                           cwd=here, env=env, capture_output=True, text=True, timeout=timeout)
    except Exception as e:      # a secondary figure must never cost the headline line
        return {'error': repr(e)[:300]}
    if r.returncode != 0:
        return {'error': (r.stderr or r.stdout)[-300:]}
    try:
        ms = {l.split(':')[0].strip(): float(l.split(':')[1].split('ms/step')[0]) for l in r.stdout.splitlines() if 'ms/step' in l}
        for l in r.stdout.splitlines():
            if l.startswith('statistic all-reduces captured in the graph:'):
                ms['collectives'] = int(l.split(':')[1])
            if l.startswith('per-shape modules:'):
                ms['per_shape_modules'] = l.split(':', 1)[1].strip()
    except (ValueError, IndexError) as e:
        return {'error': f'unparsable bench_train output: {e!r}'[:300]}
    ms['rc'] = 0
    return ms


def train_step_record(world=1, rank=0, local_rank=0, dist=None, dev=None, backend='nccl', share_device=False, steps=20):
    """The airplane config's whole training step (encoder + posterior + prior flow + 4 decoders with batch-statistic BatchNorm +
    mixture NLL + backward in one hipGraph, fused AMSGrad), timed by tools/bench_train.py in child processes after the headline
    measurement.  Secondary figures: never part of `value`.
      one rank  : the plain step, the reference's own list-API call, and the DATA-PARALLEL code path on a 1-rank RCCL group
                  (GWTF_FORCE_SHARDED=1: phase-split pipeline, 132 statistic all-reduces + row gathers + gradient exchange captured
                  in the graph) -- what that path costs before any link latency;
      N > 1     : every rank starts one child (same GPU), the children form their own RCCL group and run the sharded step:
                  `global_batch_64` = the reference's run (train_ae.py:77-78 divides the batch of 64 over the ranks) and
                  `per_rank_batch_64` (weak scaling).  ms per step = the slowest rank's."""
    base_env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    for k in list(base_env):
        # the launcher's agent settings must not reach the children: with TORCHELASTIC_USE_AGENT_STORE=True a child rank 0 would
        # expect the AGENT to host the rendezvous store on its (new) port and never start one -- every child then waits forever
        if k.startswith('TORCHELASTIC_') or k in ('GROUP_RANK', 'ROLE_RANK', 'ROLE_NAME', 'ROLE_WORLD_SIZE', 'GROUP_WORLD_SIZE',
                                                  'LOCAL_WORLD_SIZE', 'TORCH_NCCL_ASYNC_ERROR_HANDLING'):
            base_env.pop(k)
    wl = 'airplane config, whole model: forward + backward (one hipGraph) + fused AMSGrad, K=4, N=2048'
    if world == 1:
        for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'GWTF_FORCE_SHARDED'):
            base_env.pop(k, None)
        ms = _run_bench_train(['--api', 'both'], base_env, steps=steps)
        if 'hipGraph' not in ms:
            return ms if 'error' in ms else {'error': 'no hipGraph line'}
        rec = {'ms_per_step': ms['hipGraph'], 'eager_ms_per_step': ms.get('eager'), 'steps': steps, 'per_shape_modules': ms.get('per_shape_modules'),
               'list_api_ms_per_step': ms.get('hipGraph list API'),        # the reference's own call: model(g, p) -> lists -> loss
               'workload': wl + ', B=64'}
        # The literal drop-in: ONLY the decoder import swapped -- the K decoders called one at a time
        # (flow_mixture.py:163-166); `with_this_loss`: the loss import swapped too (lists -> the fused NLL kernel); `reference_loss_loop`:
        # the reference's own B x K Python loss loop on the returned lists (losses.py:109-131, restated in tools/bench_train.py).
        # `no_sibling_batching`: the same call with decoders._SiblingGroup off (K separate K = 1 pipelines: what round 4 shipped).
        sw = _run_bench_train(['--api', 'swap'], base_env, steps=steps)
        sw_off = _run_bench_train(['--api', 'swap'], dict(base_env, GWTF_NO_SIBLING_BATCH='1'), steps=max(3, steps // 4))
        sw_loop = _run_bench_train(['--api', 'swap_loop'], base_env, steps=3, timeout=300)
        rec['import_swap_only'] = {
            'with_this_loss': {'ms_per_step': sw.get('hipGraph'), 'eager_ms_per_step': sw.get('eager'), 'error': sw.get('error')},
            'with_this_loss_no_sibling_batching': {'ms_per_step': sw_off.get('hipGraph'), 'eager_ms_per_step': sw_off.get('eager'), 'error': sw_off.get('error')},
            'reference_loss_loop': {'ms_per_step': sw_loop.get('hipGraph'), 'eager_ms_per_step': sw_loop.get('eager'), 'error': sw_loop.get('error')},
            'note': 'K sequential LocalCondRNVPDecoder.forward calls on one (p, g) land in ONE K-batched pipeline pass (first call of a '
                    'ModuleList sibling group runs all K, later calls return their slices); the loss loop re-adds 34 full tensors B x K times'}
        sh = _run_bench_train([], dict(base_env, GWTF_FORCE_SHARDED='1'), steps=steps)
        rec['data_parallel_path_1rank'] = ({'ms_per_step': sh.get('hipGraph'), 'statistic_all_reduces_in_graph': sh.get('collectives'),
                                            'per_shape_modules': sh.get('per_shape_modules'),
                                            'note': 'the code path an N > 1 group runs -- SyncBatchNorm model, phase-split pipeline, one compaction launch '
                                                    'per statistic collective, row gathers, overlapped gradient exchange, every collective captured in the '
                                                    'hipGraph -- on an RCCL 1-rank group: what that path costs before any link latency'} if 'hipGraph' in sh else sh)
        # BASELINE configs[2] (autoencoding, global batch 128 over 8 GPUs): ONE rank's share, 16 shapes, G = 512, f = 33, through the same
        # data-parallel code path (1-rank RCCL group) and plain
        ae = _run_bench_train(['--config', 'ae', '--batch', '16'], dict(base_env, GWTF_FORCE_SHARDED='1'), steps=steps)
        ae_plain = _run_bench_train(['--config', 'ae', '--batch', '16'], base_env, steps=steps)
        rec['ae_shard'] = ({'workload': 'config_autoencoding.yaml per-rank shard of an 8-GPU run: K=4 x 33 couplings, f=33, G=512, 16 shapes x 2048 points, '
                                        'whole model, forward + backward in one hipGraph + fused AMSGrad',
                            'data_parallel_path_1rank_ms_per_step': ae.get('hipGraph'), 'plain_ms_per_step': ae_plain.get('hipGraph'),
                            'statistic_all_reduces_in_graph': ae.get('collectives'), 'per_shape_modules': ae.get('per_shape_modules')}
                           if 'hipGraph' in ae else ae)
        return rec
    # N > 1: a fresh rendezvous port for the children, agreed on through the parents' group
    import socket
    ports = [0, 0]
    if rank == 0:
        with socket.socket() as s1, socket.socket() as s2:
            s1.bind(('127.0.0.1', 0))
            s2.bind(('127.0.0.1', 0))
            ports = [s1.getsockname()[1], s2.getsockname()[1]]
    dist.broadcast_object_list(ports, src=0)
    env = dict(base_env, RANK=str(rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1')
    env.pop('GWTF_FORCE_SHARDED', None)
    rec = {'workload': wl + f', {world} ranks, SyncBatchNorm + overlapped gradient all-reduce, all collectives inside the graph',
           'steps': steps}
    for label, per_rank, port in (('global_batch_64', max(2, 64 // world), ports[0]), ('per_rank_batch_64', 64, ports[1])):
        # rehearsal on one GPU (--backend gloo --share-device): gloo's collectives are host-side, so no graph -- the eager step only
        graph = backend == 'nccl'
        key = 'hipGraph' if graph else 'eager'
        ms = _run_bench_train(['--batch', str(per_rank), '--backend', backend] + (['--share-device'] if share_device else []),
                              dict(env, MASTER_PORT=str(port)), timeout=150, graph=graph, steps=steps)   # a hang must not cost the headline line
        ok = torch.tensor([1.0 if ms.get('rc') == 0 and (rank != 0 or key in ms) else 0.0], device=dev if graph else 'cpu')
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)            # also keeps the parents in step between the two runs
        if float(ok) == 0.0:
            rec[label] = {'error': ms.get('error', 'a rank failed')}
            break
        if rank == 0:
            rec[label] = {'ms_per_step': ms[key], 'eager_ms_per_step': ms.get('eager'), 'per_rank_batch': per_rank,
                          'statistic_all_reduces_in_graph': ms.get('collectives'), 'graphed': graph, 'per_shape_modules': ms.get('per_shape_modules'),
                          'shapes_per_s': round(per_rank * world / ms[key] * 1e3, 1)}
    return rec


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks (one process per GPU) through torch.distributed.run,
    as the reference's train_ae.py:183-193 spawns its own workers.  The parent never touches the GPU (no HIP call before
    or after the children start); the children's rank 0 prints the JSON line on the inherited stdout."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + argv
    if args.dry_run_spawn:
        print(json.dumps({'spawn': cmd}), flush=True)
        return 0
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    return subprocess.run(cmd, env=env).returncode


def run_workload(name, args, dev, rank, world, sync_all, reduce_max):
    """Build one workload, time `args.steps` steps of it after `args.warmup` warm-up steps; -> dict of measurements."""
    cfg = WORKLOADS[name]
    K, L, f, G, B, N, mode = (cfg[k] for k in ('K', 'L', 'f', 'G', 'B', 'N', 'mode'))
    decoders = []
    for k in range(K):
        d = gw.LocalCondRNVPDecoder(L, f, G)
        load_synth_(d, 2 + k)
        decoders.append(d.to(dev).eval())
    p, g = synth_inputs(B, N, G, 1000 * rank)          # every rank owns its own shard of shapes
    pd, gd = torch.from_numpy(p).to(dev), torch.from_numpy(g).to(dev)
    eps = decoders[0].flows[0].nvp1._eps_value
    sideways = mode == 'direct' and K > 1                # sampling: each point visits ONE component
    stack = gw.MixtureStack(decoders)                    # K components: one FiLM launch + one stack launch
    counts = [N // K] * K

    # density workloads with K > 1 end in the fused mixture NLL, as the training loss consumes them (shape M2):
    # synthetic base Gaussians and mixture logits, resident like the inputs
    with_nll = (not sideways) and K > 1
    if with_nll:
        rng = np.random.default_rng(77 + rank)
        mu0 = torch.from_numpy((0.05 * rng.standard_normal((K, B, 3))).astype(np.float32)).to(dev)
        lv0 = torch.from_numpy((0.2 * rng.standard_normal((K, B, 3))).astype(np.float32)).to(dev)
        logits = torch.from_numpy(rng.standard_normal((B, K)).astype(np.float32)).to(dev)

    def launch_step():
        if sideways:
            return stack.forward_partition(pd, gd, counts, mode)
        z, ld = stack.forward_all(pd, gd, mode)
        if with_nll:
            return z, ld, _lib.mixture_nll(z, ld, mu0, lv0, logits)
        return z, ld

    probe_out = [torch.empty((B, 3, N) if sideways else (K, B, 3, N), device=dev) for _ in range(2)]

    def step(timers=None):
        """Eager path with HIP events around the stack launch ALONE (kernel-duration probe: results go to preallocated tensors, so
        nothing but the one kernel sits between the two events)."""
        pw, film, _ = stack._film(gd)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if sideways:
            segs = [(k * (N // K), (k + 1) * (N // K)) for k in range(K)]
            res = _lib.stack_forward_multi(pd, pw, film, K, stack.C, f, 0, eps, mode, segments=segs, shared_points=False,
                                           out=probe_out[0], logdet=probe_out[1])
        else:
            res = _lib.stack_forward_multi(pd, pw, film, K, stack.C, f, 0, eps, mode, out=probe_out[0], logdet=probe_out[1])
        e1.record()
        if timers is not None:
            timers.append((e0, e1))
        if with_nll:
            res = (*res, _lib.mixture_nll(res[0], res[1], mu0, lv0, logits))
        return res

    eager_step = step
    graph = None
    if not args.eager:
        # one hipGraph per step, bound to the resident p / g
        with torch.no_grad():
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                launch_step()
            torch.cuda.current_stream(dev).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with graph_capture(graph):
                graph_out = launch_step()

        def step(timers=None):                      # noqa: F811
            if timers is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                graph.replay()
                e1.record()
                timers.append((e0, e1))
            else:
                graph.replay()
            return graph_out
    else:
        def step(timers=None):                      # noqa: F811
            return eager_step(timers) if timers is not None else launch_step()

    probe = []
    if graph is not None:
        # per-kernel duration of the dominant kernel: same launches, same inputs, eager, HIP events around each stack launch
        # on the launch stream (a graph replay cannot be bracketed per kernel).  Runs BEFORE the warm-up and the timed
        # region, so the timed steps also start from settled clocks whatever --warmup is.
        with torch.no_grad():
            for _ in range(100):            # ~50 ms of load: clocks settle before anything is measured
                eager_step()
            for _ in range(max(10, args.steps // 2)):
                eager_step(probe)
        torch.cuda.synchronize(dev)

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        timers = []
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            # graph mode: no event records inside the timed region (the per-kernel probe ran before, eagerly)
            step(timers if graph is None else None)
        sync_all()
        elapsed = time.perf_counter() - t0
    elapsed = reduce_max(elapsed)

    comp_passes = 1 if sideways else K
    pts_per_launch = B * N * comp_passes                                         # all components in one launch
    if graph is not None:
        timers = probe
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in timers]))           # one stack launch, HIP events
    return dict(cfg=cfg, value=world * pts_per_launch * args.steps / elapsed / 1e6, elapsed=elapsed, kern_ms=kern_ms,
                pts_per_launch=pts_per_launch, with_nll=with_nll,
                achieved=flops_per_point(L, f) * pts_per_launch / (kern_ms * 1e-3) / 1e12)


def exact_fp32_record(name, m, dev, reps=10):
    """The same stack launch on the EXACT-fp32 contraction body (csrc/gwtf_stack_exact.hip: v_mfma_f32_16x16x4_f32, unsplit operands)
    -- the honest comparison point for the split-f16 kernel's rate: what the unit whose dtype the result has delivers on this
    workload -- and the cost of the re-run launch that follows every eval stack launch (its workgroups read their tile's results and
    leave: no point of the bench grids is out of range).  HIP events around the launches alone, results in preallocated tensors."""
    cfg = WORKLOADS[name]
    K, L, f, G, B, N, mode = (cfg[k] for k in ('K', 'L', 'f', 'G', 'B', 'N', 'mode'))
    decoders = []
    for k in range(K):
        d = gw.LocalCondRNVPDecoder(L, f, G)
        load_synth_(d, 2 + k)
        decoders.append(d.to(dev).eval())
    p, g = synth_inputs(B, N, G, 0)
    pd, gd = torch.from_numpy(p).to(dev), torch.from_numpy(g).to(dev)
    stack = gw.MixtureStack(decoders)
    eps = decoders[0].flows[0].nvp1._eps_value
    outs = [torch.empty(K, B, 3, N, device=dev) for _ in range(2)]

    def timed(fns, rounds):
        """median HIP-event duration of each fn, the fns taken in turn `rounds` times (clock drift hits all of them alike)"""
        ts = [[] for _ in fns]
        with torch.no_grad():
            for _ in range(rounds):
                for i, fn in enumerate(fns):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda._sleep(400000)     # ~0.2 ms of device-side spin: the launches below are queued before the first event fires
                    e0.record()
                    fn()
                    e1.record()
                    ts[i].append((e0, e1))
        torch.cuda.synchronize(dev)
        return [float(np.median([a.elapsed_time(b) for a, b in t])) for t in ts]
    with torch.no_grad():
        pw, film, _ = stack._film(gd)
        px = stack.packed_exact()
    split = lambda x: _lib.stack_forward_multi(pd, pw, film, K, stack.C, f, 0, eps, mode, out=outs[0], logdet=outs[1], packed_x=x)

    def exact():
        with _lib.exact_fp32():
            split(px)
    with torch.no_grad():
        for _ in range(100):                  # ~50 ms of load: clocks settle before anything is measured
            split(px)
    t_split, _, t_exact = timed([lambda: split(None), lambda: split(px), exact], reps)
    # the re-run launch's cost where it matters: inside a hipGraph (one graph of the split launch alone, one with the re-run behind it)
    graphs = []
    with torch.no_grad():
        for x in (None, px):
            gr = torch.cuda.CUDAGraph()
            with graph_capture(gr):
                for _ in range(10):
                    split(x)
            graphs.append(gr)
    t_g = timed([graphs[0].replay, graphs[1].replay], reps)
    t_both = t_split + (t_g[1] - t_g[0]) / 10
    tf = flops_per_point(L, f) * K * B * N / (t_exact * 1e-3) / 1e12
    return {'workload': cfg['name'], 'kernel': 'stack_exact_kernel (v_mfma_f32_16x16x4_f32, unsplit fp32 operands)',
            'kernel_ms': round(t_exact, 4), 'achieved': round(tf, 2), 'unit': 'TFLOP/s', 'peak_fp32_mfma': MFMA_F32_PEAK_TFLOPS,
            'frac_of_fp32_mfma_peak': round(tf / MFMA_F32_PEAK_TFLOPS, 4), 'split_f16_kernel_ms': round(t_split, 4),
            'split_over_exact': round(t_exact / t_split, 2), 'rerun_launch_us': round((t_both - t_split) * 1e3, 2),
            'note': 'rerun_launch_us: the exact-fp32 re-run launch behind the split launch when no tile is flagged, measured inside a hipGraph (10 launches with it - 10 without) / 10; it is part of every timed step above'}


def under_profiler():
    """True when this process already runs under rocprofv3 (its tool library is preloaded / its environment is set)."""
    return 'rocprof' in os.environ.get('LD_PRELOAD', '').lower() or any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ)


def live_traffic(workload, timeout=150):
    """HBM-side bytes per stack_kernel launch measured DURING this run: two child runs of this script under
    `rocprofv3 --kernel-trace --pmc <counter>` (FETCH_SIZE, then WRITE_SIZE: separate passes, eager launches, 5 steps), corrected
    as MI355X_MICROARCH.md's HBM section prescribes (gfx950: read bytes = 2 x FETCH_SIZE x 1024, write = WRITE_SIZE x 1024).
    -> (bytes or None, source text).  The program after `--` is the interpreter itself (no launcher in between)."""
    import csv, glob, shutil, tempfile
    exe = shutil.which('rocprofv3')
    if exe is None:
        return None, 'rocprofv3 not on PATH'
    vals = {}
    for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix='gwtf_pmc_', dir='/tmp')
        cmd = [exe, '--kernel-trace', '--pmc', ctr, '--output-format', 'csv', '-d', d, '-o', 't', '--', sys.executable,
               os.path.abspath(__file__), '--workload', workload, '--no-cpu-baseline', '--no-also', '--eager', '--steps', '5', '--warmup', '2',
               '--no-live-traffic']
        try:
            subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           timeout=timeout, check=True)
            got = []
            for fn in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
                for r in csv.DictReader(open(fn)):
                    if 'stack_kernel' in r['Kernel_Name'] and r['Counter_Name'] == ctr:
                        got.append(float(r['Counter_Value']))
            if not got:
                return None, f'rocprofv3 --pmc {ctr}: no stack_kernel rows'
            vals[ctr] = sum(got) / len(got)
        except Exception as e:  # noqa: BLE001 -- any failure of the profiler child leaves the static figure in place
            return None, f'rocprofv3 --pmc {ctr} child failed: {type(e).__name__}'
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return int(round(2 * vals['FETCH_SIZE'] * 1024 + vals['WRITE_SIZE'] * 1024, -3)), \
        'measured by this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE child passes of bench.py (eager, 5 steps), mean per stack_kernel launch, 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024'


def roofline_record(name, m, live=False):
    """`roofline` object of the JSON line.  The unit that executes the f x f contraction is the f16 matrix pipe (three
    f16 MFMA products per fp32 product, fp32 accumulate): its dense peak / 3 is the fp32-equivalent roof `frac` is
    quoted against.  The fp32-MFMA roof (the dtype of the result) is kept as a secondary pair; against it the ratio can
    exceed 1 because that unit is not the one doing the work."""
    cfg = m['cfg']
    peak = MFMA_F16_PEAK_TFLOPS / 3
    # HBM-side bytes per launch, measured in this run (live_traffic: two rocprofv3 --pmc child passes) for the headline and the metric's
    # own shape; null where not measured -- no figure is carried over from an earlier profile
    traffic, traffic_source = None, None
    if live:
        traffic, traffic_source = live_traffic(name)
        if traffic is None:
            traffic_source = 'not measured: ' + str(traffic_source)
    return {'bound': 'mfma', 'achieved': round(m['achieved'], 3), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
            'frac': round(m['achieved'] / peak, 4), 'traffic': traffic, 'traffic_source': traffic_source,
            'peak_basis': 'dense f16 MFMA 2500 TFLOP/s / 3 products per fp32 product (the executing unit)',
            'peak_fp32_mfma': MFMA_F32_PEAK_TFLOPS, 'ratio_vs_fp32_mfma': round(m['achieved'] / MFMA_F32_PEAK_TFLOPS, 4),
            'kernel': 'stack_kernel (fused coupling stack)', 'kernel_ms': round(m['kern_ms'], 4),
            'flop_per_point': flops_per_point(cfg['L'], cfg['f']), 'points_per_launch': m['pts_per_launch']}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=100)
    ap.add_argument('--workload', default='airplane', choices=sorted(WORKLOADS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-live-traffic', action='store_true', help='skip the rocprofv3 --pmc child passes that measure roofline.traffic (it is then null)')
    ap.add_argument('--no-also', action='store_true', help='skip the secondary M1 (north-star shape) measurement')
    ap.add_argument('--no-train-step', action='store_true', help='skip the secondary whole-model training-step measurement')
    ap.add_argument('--train-step-steps', type=int, default=20, help='timed steps of the secondary training-step measurement')
    ap.add_argument('--also-select', default='m1,ae,svr,k16,k16_b1,exact_fp32,train_step',
                    help='comma list of the secondary measurements to run beside the airplane headline')
    ap.add_argument('--eager', action='store_true', help='launch through the eager module path instead of one hipGraph per step')
    ap.add_argument('--points-per-wave', type=int, default=0, help='tuning hook: 16/32/64, 0 = library default')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='process-group backend for the barrier / max-over-ranks (nccl = RCCL; gloo: rehearsal on one GPU)')
    ap.add_argument('--share-device', action='store_true',
                    help='rehearsal: every rank uses cuda:0 (RCCL refuses two ranks on one device, so use --backend gloo)')
    ap.add_argument('--lib', default=None, help='tuning hook: load this build of libgwtf_hip.so (A/B of two builds in one call)')
    ap.add_argument('--dry-run-spawn', action='store_true', help='print the launcher command instead of running it')
    args = ap.parse_args()
    if args.lib:
        _lib.LIB_PATH = args.lib

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args, sys.argv[1:]))        # before any GPU call in this process
    args.gpus = world
    dev_index = 0 if args.share_device else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)   # RCCL; used only for the barrier + max-over-ranks
        else:
            dist.init_process_group('gloo')
    _lib.set_tuning(args.points_per_wave)

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def reduce_max(x):
        if dist is None:
            return x
        t = torch.tensor([x], device=dev if args.backend == 'nccl' else 'cpu', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    m = run_workload(args.workload, args, dev, rank, world, sync_all, reduce_max)
    also = None
    if args.workload == 'airplane' and not args.no_also:
        # the metric's own shape (B=32 x N=2048, the north-star module) and the other BASELINE configs' shapes, timed by the
        # same protocol in the same run (a few seconds each)
        also = {}
        selected = [n for n in args.also_select.split(',') if n]
        for name in [n for n in ('m1', 'ae', 'svr', 'k16', 'k16_b1') if n in selected]:
            a = run_workload(name, args, dev, rank, world, sync_all, reduce_max)
            also[name] = {'value': round(a['value'], 3), 'unit': 'Mpoints/s',
                          'ms_per_step': round(a['elapsed'] / args.steps * 1e3, 4), 'steps': args.steps, 'warmup': args.warmup,
                          'workload': a['cfg']['name'],
                          # every shape's HBM traffic is measured by this run (two rocprofv3 --pmc child passes each, a few seconds)
                          'roofline': roofline_record(name, a, live=(world == 1 and not args.no_live_traffic and not under_profiler()))}

        if rank == 0 and 'exact_fp32' in selected:
            also['exact_fp32'] = exact_fp32_record('airplane', m, dev)
        if not args.no_train_step and 'train_step' in selected:
            ts = train_step_record(world, rank, local_rank, dist, dev, args.backend, args.share_device, args.train_step_steps)      # N > 1: every rank takes part (one child per GPU)
            if rank == 0:
                also['train_step'] = ts

    if rank == 0:
        cfg = m['cfg']
        K, L, f, G, B, N, mode = (cfg[k] for k in ('K', 'L', 'f', 'G', 'B', 'N', 'mode'))
        line = {
            'metric': 'point-flow fwd+logdet Mpoints/sec (B x 2048 pts)', 'value': round(m['value'], 3), 'unit': 'Mpoints/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(m['elapsed'] / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
            # I/O, accumulators and every elementwise step are fp32; the f x f contraction is three f16 MFMA products of hi / lo split
            # fp32 operands (fp32-grade: DESIGN.md section 5); GWTF_TUNE_EXACT_FP32 runs it on the fp32 MFMA instead (also.exact_fp32)
            'vs_baseline': None, 'dtype': 'f32 (split-f16 x3 MFMA, fp32 accumulate)', 'data': 'synthetic',
            'config': {'workload': cfg['name'] + (' -- leads the line because it is the largest single-GPU config of BASELINE.json '
                                                  '(configs[1]); the north-star shape B=32 x 2048, f=64 is also.m1' if args.workload == 'airplane' else ''),
                       'per_gpu_batch': B, 'points_per_shape': N, 'components': K,
                       'couplings_per_component': 3 * L, 'f': f, 'G': G, 'direction': mode,
                       'point_definition': 'one 3-D point through one component stack (coords + sum logvars)',
                       'sharding': f'batch of shapes over {world} rank(s), no data-path collective',
                       'launch': ('eager: ' if args.eager else 'one hipGraph replay per step: ') + '1 FiLM + 1 stack launch for all components'
                                 + (' + 1 mixture-NLL launch (per-shape NLL over K components)' if m['with_nll'] else '')},
            # the default invocation measures the headline kernel's HBM traffic itself (never from inside another profiler)
            'roofline': roofline_record(args.workload, m, live=(world == 1 and not args.no_live_traffic and also is not None and not under_profiler())),
        }
        if also is not None:
            line['also'] = also
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(cfg)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
