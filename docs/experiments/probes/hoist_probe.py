"""Where do the generic and the pipelined coupling body differ (a build given by argv[1])?  f = 37, 64 x 2048 points."""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from go_with_the_flows_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = sys.argv[1]
import go_with_the_flows_amd as gw
from test_gpu_parity import decoder_and_state, synth_inputs, dev, DEV
for (B, N) in ((64, 2048), (4, 256), (64, 256), (4, 2048)):
    m, _ = decoder_and_state(1, 37, 32, 77); m = m.to(DEV).eval()
    p, g = synth_inputs(B, N, 32, 78); pd, gd = dev(p), dev(g)
    outs = []
    for flag in (0, 1 << 30):
        _lib.set_tuning(flag)
        with torch.no_grad():
            ps, mus, lvs = m(pd, gd, mode='direct')
        outs.append(torch.stack(mus))
    _lib.set_tuning(0)
    e = (outs[0] - outs[1]).abs()                       # (C, B, 3, N)
    bad = (e.amax(dim=(0, 2)) > 1e-5)                   # (B, N)
    nb = bad.nonzero()
    print(B, N, 'max', float(e.max()), 'bad points', int(bad.sum()), 'per coupling max', [float(x) for x in e.amax(dim=(1, 2, 3))])
    if len(nb):
        ns = nb[:, 1]
        print('   n % 256 histogram (16-bins):', torch.bincount((ns % 256) // 16, minlength=16).tolist(), ' shapes:', sorted(set(nb[:, 0].tolist()))[:10])
