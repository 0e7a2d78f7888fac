"""Run by tests/test_gpu_parity.py in a subprocess: the flat gradient all-reduce over RCCL (backend 'nccl') on a 1-rank
group -- the only RCCL configuration one GPU allows (RCCL refuses two ranks on one device)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import go_with_the_flows_amd as gw                                   # noqa: E402
from go_with_the_flows_amd import dist as gdist                      # noqa: E402
from go_with_the_flows_amd.synth import load_synth_, synth_inputs    # noqa: E402

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', str(29700 + os.getpid() % 200))
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
m = gw.LocalCondRNVPDecoder(1, 19, 16)
load_synth_(m, 3)
m = m.to(dev).train()
p, g = synth_inputs(4, 128, 16, 5)
z, ld = m.forward_fused(torch.from_numpy(p).to(dev), torch.from_numpy(g).to(dev), 'inverse')
(0.5 * (ld + z * z).sum() / 4).backward()
before = [q.grad.clone() for q in m.parameters()]
n = gdist.all_reduce_gradients(m, average=True, force=True)
torch.cuda.synchronize()
assert n == sum(q.numel() for q in m.parameters()), n
assert all(torch.equal(a, q.grad) for a, q in zip(before, m.parameters()))
# the phase-split train pipeline with its statistic all-reduces going over RCCL (1-rank group): same numbers as the
# single-call pipeline, 4 C collectives
from go_with_the_flows_amd import autograd as gwa                   # noqa: E402
os.environ['GWTF_FORCE_SHARDED'] = '1'
gwa.COLLECTIVES['n'] = 0
m.zero_grad(set_to_none=True)
z2, ld2 = m.forward_fused(torch.from_numpy(p).to(dev), torch.from_numpy(g).to(dev), 'inverse')
(0.5 * (ld2 + z2 * z2).sum() / 4).backward()
torch.cuda.synchronize()
assert gwa.COLLECTIVES['n'] == 4 * 3, gwa.COLLECTIVES
assert float((z2 - z).abs().max()) < 1e-5 and float((ld2 - ld).abs().max()) < 1e-5
worst = max(float((a - q.grad).abs().max() / (a.abs().max() + 1e-12)) for a, q in zip(before, m.parameters()))
assert worst < 1e-4, worst
del os.environ['GWTF_FORCE_SHARDED']
t = torch.full((1 << 20,), 2.0, device=dev)
dist.all_reduce(t)
assert float(t.sum()) == 2.0 * (1 << 20)
assert gdist.max_over_ranks(1.25, dev) == 1.25
dist.barrier()
dist.destroy_process_group()
print('RCCL1 ok', n)
