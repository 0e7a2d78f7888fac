"""Who launches the fill / copy kernels of the end-to-end training step?  (torch profiler, grouped by python stack)"""
import sys, runpy, torch
sys.argv = ['bench_train.py', '--steps', '1']
sys.path.insert(0, '.')
ns = runpy.run_path('tools/bench_train.py', run_name='notmain')
model, crit, opt, g_in, p_in = ns['model'], ns['crit'], ns['opt'], ns['g_in'], ns['p_in']
def step():
    opt.zero_grad(set_to_none=True)
    enc, dec = model.forward_fused(g_in, p_in)
    loss = crit.fused(enc, dec)[0]
    loss.backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    step()
torch.cuda.synchronize()
import collections
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ('aten::fill_', 'aten::zero_', 'aten::copy_', 'aten::zeros', 'aten::clone', 'aten::contiguous', 'aten::add_', 'aten::add', 'aten::mul', 'aten::cat'):
        st = [s for s in ev.stack if 'go_with_the_flows_amd' in s or 'bench_train' in s or 'autograd' in s][:2]
        cnt[(ev.name, tuple(st))] += 1
for (name, st), n in cnt.most_common(40):
    print(n, name, ' <- '.join(s.split('/')[-1] for s in st))
