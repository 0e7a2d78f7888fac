"""Device kernels of one eager training step that are NOT the pipelines' own: by autograd node / module, with durations."""
import sys, runpy, collections, torch
sys.argv = ['bench_train.py', '--steps', '1']
sys.path.insert(0, '.')
ns = runpy.run_path('tools/bench_train.py', run_name='notmain')
step = ns['step']
from torch.profiler import profile, ProfilerActivity
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
evs = prof.events()
# map each device kernel to the top-level CPU op (autograd node or aten op directly under the profiler root) that launched it
cpu = [e for e in evs if e.device_type == torch.autograd.DeviceType.CPU]
def top(e):
    while e.cpu_parent is not None and e.cpu_parent.cpu_parent is not None:
        e = e.cpu_parent
    return e
agg = collections.defaultdict(lambda: [0, 0.0])
for e in cpu:
    ks = [k for k in e.kernels] if hasattr(e, 'kernels') else []
    if not ks or any(c for c in e.cpu_children if getattr(c, 'kernels', None)):
        continue
    t = top(e)
    skip = ('TrainMixtureFn', '_PriorFlowFn', '_EncoderTrainFn', 'NLL')
    if any(s in t.name for s in skip):
        continue
    a = agg[(t.name[:60], e.name[:40])]
    a[0] += len(ks); a[1] += sum(k.duration for k in ks)
tot = sum(v[1] for v in agg.values())
print('glue kernels: %d launches, %.0f us of device time' % (sum(v[0] for v in agg.values()), tot))
for (tn, en), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print('%7.1f us %3d  %-40s <- %s' % (us, n, en, tn))
