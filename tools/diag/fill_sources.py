"""Where do the fill kernels of one airplane training step come from?  (torch profiler, Python stacks of aten::fill_/zero_)"""
import collections, sys, runpy, torch
sys.argv = ['bench_train.py', '--steps', '1']
ns = runpy.run_path('tools/bench_train.py', run_name='not_main') if False else None
sys.path.insert(0, '.')
import importlib.util
spec = importlib.util.spec_from_file_location('bt', 'tools/bench_train.py')
bt = importlib.util.module_from_spec(spec); spec.loader.exec_module(bt)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    bt.step()
torch.cuda.synchronize()
cnt = collections.Counter(); other = collections.Counter()
for e in prof.events():
    if e.name in ('aten::fill_', 'aten::zero_', 'aten::zeros', 'aten::zeros_like', 'aten::new_zeros'):
        st = [s for s in e.stack if 'site-packages' not in s and 'dist-packages' not in s][:3]
        cnt[(e.name, ' <- '.join(st) if st else ' | '.join(e.stack[:4]))] += 1
for k, v in cnt.most_common(25):
    print(v, k)
