"""Probe: which pattern makes ProcessGroupNCCL's watchdog query a captured event (hipErrorCapturedEvent)?"""
import os
import sys
import time

import torch
import torch.distributed as dist

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', str(29800 + os.getpid() % 100))
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
which = sys.argv[1]
x = torch.ones(1 << 16, device=dev)
side2 = torch.cuda.Stream()


def body():
    y = x * 2
    dist.all_reduce(y)
    if which in ('side', 'side_norecord'):
        cur = torch.cuda.current_stream()
        buf = y.clone()
        side2.wait_stream(cur)
        with torch.cuda.stream(side2):
            dist.all_reduce(buf)
        if which == 'side':
            buf.record_stream(side2)
        z = y + 1
        cur.wait_stream(side2)
        return z + buf
    if which == 'thread':
        # a collective issued from the autograd thread (backward hook)
        w = torch.ones(4, device=dev, requires_grad=True)
        out = (w * y[:4]).sum()
        w.register_hook(lambda g: dist.all_reduce(g.clone()) and None)
        out.backward()
        return w.grad
    return y


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        body()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode='thread_local'):
    out = body()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
time.sleep(2.0)
dist.all_reduce(x)
torch.cuda.synchronize()
time.sleep(1.0)
print(which, 'OK', float(out.reshape(-1)[0]))
dist.destroy_process_group()
