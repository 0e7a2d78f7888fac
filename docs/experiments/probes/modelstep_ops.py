"""Which torch glue kernels does the end-to-end training step launch, and from which part?  (torch profiler, device activity)"""
import sys, runpy, collections, torch
sys.argv = ['bench_train.py', '--steps', '1']
sys.path.insert(0, '.')
ns = runpy.run_path('tools/bench_train.py', run_name='notmain')
model, crit, opt, g_in, p_in = ns['model'], ns['crit'], ns['opt'], ns['g_in'], ns['p_in']
from torch.profiler import profile, ProfilerActivity, record_function

def step():
    opt.zero_grad(set_to_none=True)
    with record_function('R_forward'):
        enc, dec = model.forward_fused(g_in, p_in)
        loss = crit.fused(enc, dec)[0]
    with record_function('R_backward'):
        loss.backward()
    with record_function('R_opt'):
        opt.step()

for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
evs = prof.events()
regions = [(e.name, e.time_range.start, e.time_range.end) for e in evs if e.name.startswith('R_')]
cnt = collections.Counter(); tim = collections.Counter()
for e in evs:
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith('aten::') and e.cpu_parent is not None and not e.cpu_parent.name.startswith('aten::'):
        reg = next((r[0] for r in regions if r[1] <= e.time_range.start <= r[2]), '?')
        cnt[(reg, e.name, e.cpu_parent.name[:40])] += 1
for (reg, name, par), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:45]:
    print(n, reg, name, '<-', par)
