"""HBM roofline check of the fused Adam kernel: bytes per element 28 (Adam) / 36 (AMSGrad)."""
import sys, torch
sys.path.insert(0, ".")
from go_with_the_flows_amd.optim import Adam
for n_tensors, numel in [(1, 64 * 1024 * 1024), (1500, 17320), (342 * 4, 3500)]:
    for ams in (False, True):
        ps = [torch.nn.Parameter(torch.randn(numel, device="cuda")) for _ in range(n_tensors)]
        for q in ps: q.grad = torch.randn_like(q)
        opt = Adam(ps, lr=1e-3, weight_decay=1e-4, amsgrad=ams)
        for _ in range(3): opt.step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): opt.step()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        byts = n_tensors * numel * (36 if ams else 28)
        print(f"{n_tensors:5d} tensors x {numel:9d} elems amsgrad={ams}: {ms*1e3:8.1f} us/step  {byts/ms/1e6:7.1f} GB/s  ({byts/ms/1e6/8000*100:.1f}% of 8 TB/s)")
