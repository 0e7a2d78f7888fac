"""Random sweep: pipelined vs generic coupling body over widths / batch shapes / tile sizes / directions (list outputs)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import go_with_the_flows_amd as gw
from go_with_the_flows_amd import _lib
from go_with_the_flows_amd.synth import load_synth_, synth_inputs
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
widths = [17, 18, 19, 20, 33, 34, 35, 36, 37, 38, 39, 40, 61, 62, 63, 64]
worst = 0.0; nbad = 0
for it in range(160):
    f = int(rng.choice(widths)); L = int(rng.integers(1, 4)); G = int(rng.choice([8, 24, 128]))
    B = int(rng.integers(1, 9)); N = int(rng.choice([1, 15, 16, 17, 63, 64, 65, 127, 129, 255, 257, 700, 1024, 2048, 2500]))
    mode = 'direct' if rng.random() < 0.5 else 'inverse'; ppw = int(rng.choice([0, 16, 32, 64]))
    d = gw.LocalCondRNVPDecoder(L, f, G); load_synth_(d, int(rng.integers(1, 1000))); d = d.cuda().eval()
    p, g = synth_inputs(B, N, G, int(rng.integers(1, 1000))); pd, gd = torch.from_numpy(p).cuda(), torch.from_numpy(g).cuda()
    outs = []
    for flag in (ppw, ppw | 1 << 30):
        _lib.set_tuning(flag)
        with torch.no_grad():
            ps, mus, lvs = d(pd, gd, mode=mode)
            z, ld = d.forward_fused(pd, gd, mode)
        outs.append(torch.cat([torch.stack(ps).flatten(), torch.stack(mus).flatten(), torch.stack(lvs).flatten(), z.flatten(), ld.flatten()]))
    e = float((outs[0] - outs[1]).abs().max()); sc = max(1.0, float(outs[1].abs().max()))
    worst = max(worst, e / sc)
    if not np.isfinite(e) or e > 5e-6 * sc:
        nbad += 1; print('BAD', dict(f=f, L=L, G=G, B=B, N=N, mode=mode, ppw=ppw), e, sc)
_lib.set_tuning(0)
print('cases 160, bad', nbad, 'worst relative difference', worst)
