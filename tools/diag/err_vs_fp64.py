"""Distance of the HIP path to the reference's own fp64 evaluation on the golden decoder cases (coords / per-point log-det),
next to the distance of the reference's fp32 evaluation: the yardstick for any change of the arithmetic."""
import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import os
from helpers import decoder_and_state
golden = lambda n: np.load(os.path.join('tests', 'golden', n + '.npz'))
for name in ['g3_decoder_4x64x128', 'g4_width37', 'g4_width33', 'g4_width19']:
    D = golden(name)
    L, f, G, B, N, seed = D['dims']
    m, _ = decoder_and_state(L, f, G, seed)
    m = m.cuda().eval()
    for mode in ('direct', 'inverse'):
        tag = f'eval_{mode}'
        with torch.no_grad():
            out, ld = m.forward_fused(torch.from_numpy(D['p']).cuda(), torch.from_numpy(D['g']).cuda(), mode=mode)
        ref64 = D['first64_' + tag] if mode == 'inverse' else D['last64_' + tag]
        ref32 = D['first_' + tag] if mode == 'inverse' else D['last_' + tag]
        e = np.abs(out.cpu().numpy() - ref64); e32 = np.abs(ref32 - ref64)
        el = np.abs(ld.cpu().numpy() - D['logdet64_' + tag]); el32 = np.abs(D['logdet_' + tag] - D['logdet64_' + tag])
        print(f'{name:22s} {mode:8s} coords: hip max {e.max():.2e} mean {e.mean():.2e} | ref-fp32 max {e32.max():.2e} mean {e32.mean():.2e}'
              f' || logdet: hip max {el.max():.2e} mean {el.mean():.2e} | ref-fp32 max {el32.max():.2e} mean {el32.mean():.2e}')
