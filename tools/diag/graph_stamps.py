"""Where do the parallel branches of the captured training step really run?  One-thread stamp kernels (gwtf_diag_stamp: wall_clock64,
100 MHz) before and after the prior flow's forward / backward (side stream) and the decoders' forward / backward (main stream) are
captured with the step; after a replay the stamps give the true timeline -- rocprofv3's kernel trace serialises the streams."""
import sys, runpy, ctypes, torch
batch = sys.argv[1] if len(sys.argv) > 1 else '64'
sys.argv = ['bench_train.py', '--steps', '1', '--batch', batch]
sys.path.insert(0, '.')
ns = runpy.run_path('tools/bench_train.py', run_name='notmain')
from go_with_the_flows_amd import _lib, prior, autograd, encoders
stamps = torch.zeros(32, dtype=torch.int64, device='cuda')
names = {}

def stamp(name):
    i = names.setdefault(name, len(names))
    _lib.check(_lib.lib().gwtf_diag_stamp(stamps[i:].data_ptr(), _lib._stream(stamps)))

def wrap(cls, tag):
    f, b = cls.forward, cls.backward
    def fw(ctx, *a, **k):
        stamp(tag + ' fwd start'); r = f(ctx, *a, **k); stamp(tag + ' fwd end'); return r
    def bw(ctx, *a):
        stamp(tag + ' bwd start'); r = b(ctx, *a); stamp(tag + ' bwd end'); return r
    cls.forward, cls.backward = staticmethod(fw), staticmethod(bw)

wrap(prior._PriorFlowFn, 'prior')
wrap(autograd.TrainMixtureFn, 'decoders')
wrap(encoders._EncoderTrainFn, 'encoder')
fwd_bwd, opt = ns['fwd_bwd'], ns['opt']

def step():
    stamp('step start'); l = fwd_bwd(); stamp('backward done'); return l

for _ in range(3):
    step(); opt.step()
torch.cuda.synchronize()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        step()
for _ in range(5):
    g.replay(); opt.step()
torch.cuda.synchronize()
t = stamps.cpu().tolist()
t0 = t[names['step start']]
for n, i in sorted(names.items(), key=lambda kv: t[kv[1]]):
    print('%-22s %9.1f us' % (n, (t[i] - t0) / 100.0))
