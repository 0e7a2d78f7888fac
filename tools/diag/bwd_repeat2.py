import os, sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import decoder_and_state
from go_with_the_flows_amd.synth import synth_inputs
from go_with_the_flows_amd import autograd as gwa
import torch.distributed as dist
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29871')
dist.init_process_group('gloo', rank=0, world_size=1)
os.environ['GWTF_FORCE_SHARDED'] = '1'
mode = sys.argv[1]
if mode == 'sync':
    gwa._stat_sum = lambda t: torch.cuda.synchronize()
elif mode == 'nosync':
    gwa._stat_sum = lambda t: None
L, f, G, B, N = 2, 8, 16, 4, 48
p, g = synth_inputs(B, N, G, 1)
m, _ = decoder_and_state(L, f, G, 7); m = m.cuda().train()
pd, gd = torch.from_numpy(p).cuda().requires_grad_(True), torch.from_numpy(g).cuda().requires_grad_(True)
z, ld = m.forward_fused(pd, gd, 'inverse'); loss = (z * z).sum() + ld.sum()
outs = []
for i in range(6):
    gp, = torch.autograd.grad(loss, [pd], retain_graph=True)
    torch.cuda.synchronize()
    outs.append(gp.clone())
print(mode, [round(float((outs[i] - outs[0]).abs().max()), 6) for i in range(1, 6)])
