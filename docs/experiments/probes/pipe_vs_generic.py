"""Difference between the pipelined and the generic coupling body, per width and points-per-wave (mu / logvar lists)."""
import sys, torch
sys.path.insert(0, ".")
import go_with_the_flows_amd as gw
from go_with_the_flows_amd import _lib
from go_with_the_flows_amd.synth import load_synth_, synth_inputs
for f in [int(a) for a in sys.argv[1:]] or [37, 33, 40, 19]:
    d = gw.LocalCondRNVPDecoder(2, f, 32); load_synth_(d, 5); d = d.cuda().eval()
    p, g = synth_inputs(4, 256, 32, 6); pd, gd = torch.from_numpy(p).cuda(), torch.from_numpy(g).cuda()
    for ppw in (16, 32, 64):
        outs = []
        for flag in (ppw, ppw | 1 << 30):
            _lib.set_tuning(flag)
            with torch.no_grad():
                ps, mus, lvs = d(pd, gd, mode='direct')
            outs.append((torch.stack(mus), torch.stack(lvs)))
        _lib.set_tuning(0)
        for name, a, b in zip(('mu', 'lv'), outs[0], outs[1]):
            e = (a - b).abs()
            print(f, ppw, name, 'max', float(e.max()), 'coupling 3 per dim', [float(x) for x in e[3].amax(dim=(0, 2))],
                  'n_bad', int((e[3] > 1e-6).sum()), 'first bad points', (e[3].amax(dim=(0,1)) > 1e-6).nonzero().flatten()[:12].tolist())
