"""Probe (GPU box): are RCCL collectives issued through torch.distributed capturable in a hipGraph on a 1-rank nccl group, and
what does one small in-graph all-reduce cost beside dependent kernels?"""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', str(29800 + os.getpid() % 100))
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
x = torch.ones(1200, device=dev)
y = torch.zeros(1200, device=dev)


def body(n, collective):
    for _ in range(n):
        y.add_(x)                     # a dependent small kernel
        if collective:
            dist.all_reduce(y)


for coll in (False, True):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body(4, coll)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    y.zero_()
    try:
        with torch.cuda.graph(g):
            body(132, coll)
    except Exception as e:
        print('capture failed', coll, repr(e)[:400])
        continue
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f'collective={coll}: {dt * 1e3:.3f} ms per replay of 132 steps = {dt / 132 * 1e6:.2f} us per step; y[0]={float(y[0])}')
# gather into tensor + async op inside capture
z = torch.zeros(4, 1200, device=dev)
g2 = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    dist.all_gather_into_tensor(z[:1].view(-1), x)
torch.cuda.current_stream().wait_stream(s)
try:
    with torch.cuda.graph(g2):
        dist.all_gather_into_tensor(z[:1].view(-1), x)
        w = dist.all_reduce(y, async_op=True)
        y2 = x * 2
        w.wait()
        y3 = y + y2
    g2.replay(); torch.cuda.synchronize()
    print('all_gather_into_tensor + async all_reduce captured OK', float(y3[0]))
except Exception as e:
    print('capture 2 failed', repr(e)[:400])
dist.destroy_process_group()
