"""Glue kernels of one eager training step by the PYTHON LINE of this package that launched them (torch.profiler with_stack)."""
import sys, runpy, collections, torch
sys.argv = ['bench_train.py', '--steps', '1']
sys.path.insert(0, '.')
ns = runpy.run_path('tools/bench_train.py', run_name='notmain')
step = ns['step']
from torch.profiler import profile, ProfilerActivity
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU: continue
    ks = getattr(e, 'kernels', None) or []
    if not ks or any(getattr(c, 'kernels', None) for c in e.cpu_children): continue
    if any(k.name.startswith('void (anonymous namespace)') or '(anonymous namespace)::' in k.name and 'at::native' not in k.name for k in ks): continue
    frame = next((f for f in (e.stack or []) if 'go_with_the_flows_amd' in f or 'bench_train' in f), None)
    if frame is None:
        p = e.cpu_parent
        while p is not None and frame is None:
            frame = next((f for f in (p.stack or []) if 'go_with_the_flows_amd' in f or 'bench_train' in f), None)
            p = p.cpu_parent
    key = (frame or 'autograd engine / unknown').split('/')[-1][:70]
    a = agg[(key, e.name[:28])]
    a[0] += len(ks); a[1] += sum(k.duration for k in ks)
print('glue: %d launches, %.0f us' % (sum(v[0] for v in agg.values()), sum(v[1] for v in agg.values())))
for (k, n), (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print('%7.1f us %3d  %-28s %s' % (us, c, n, k))
