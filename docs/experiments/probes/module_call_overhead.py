"""Host cost per module-level call (eval): decoder.forward_fused and the list API, back to back without syncs."""
import sys, time, torch
sys.path.insert(0, ".")
import go_with_the_flows_amd as gw
from go_with_the_flows_amd.synth import load_synth_, synth_inputs
for (L, f, G, B, N) in [(4, 64, 128, 32, 2048), (11, 37, 128, 64, 2048)]:
    d = gw.LocalCondRNVPDecoder(L, f, G); load_synth_(d, 2); d = d.cuda().eval()
    p, g = synth_inputs(B, N, G, 0); pd, gd = torch.from_numpy(p).cuda(), torch.from_numpy(g).cuda()
    with torch.no_grad():
        for name, fn in (('forward_fused', lambda: d.forward_fused(pd, gd, 'inverse')), ('forward (lists)', lambda: d(pd, gd, mode='inverse'))):
            for _ in range(5): fn()
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(200): fn()
            host = (time.perf_counter() - t) / 200
            torch.cuda.synchronize(); tot = (time.perf_counter() - t) / 200
            print(f'L={L} f={f}: {name:16s} host {host*1e6:7.1f} us per call, with GPU {tot*1e6:7.1f} us')
