"""Which Python frames launch the small kernels of one airplane training step?  (torch profiler; forward ops by their stack,
backward ops by the forward frame that created the autograd node)"""
import collections, sys, torch, importlib.util
sys.argv = ['bench_train.py', '--steps', '1']
sys.path.insert(0, '.')
spec = importlib.util.spec_from_file_location('bt', 'tools/bench_train.py')
bt = importlib.util.module_from_spec(spec); spec.loader.exec_module(bt)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    bt.step()
    torch.cuda.synchronize()
by_frame = collections.Counter(); by_op = collections.Counter()
for e in prof.events():
    nk = len(e.kernels) if hasattr(e, 'kernels') else 0
    if nk == 0 or e.cpu_children:      # leaf ops that launched kernels
        continue
    fr = [s for s in e.stack if 'go_with_the_flows_amd' in s or 'bench_train' in s]
    by_frame[(fr[0].split('/')[-1] if fr else '<autograd engine / other>')] += nk
    by_op[e.name] += nk
print('kernel launches by frame:')
for k, v in by_frame.most_common(25): print('  %5d  %s' % (v, k))
print('kernel launches by op:')
for k, v in by_op.most_common(25): print('  %5d  %s' % (v, k))
