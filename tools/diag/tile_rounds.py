"""Stack-kernel time against the number of workgroups at each tile size (16 / 32 / 64 points per wave): how a partial resident
round is priced.  One component, f = 33 (MB = 3), 33 couplings; B x 2048 points with B swept."""
import sys, torch
sys.path.insert(0, ".")
import go_with_the_flows_amd as gw
from go_with_the_flows_amd import _lib
from go_with_the_flows_amd.synth import load_synth_, synth_inputs
L, f, G = 11, int(sys.argv[1]) if len(sys.argv) > 1 else 33, 128
d = gw.LocalCondRNVPDecoder(L, f, G); load_synth_(d, 2); d = d.cuda().eval()
for ppw in (64, 32, 16):
    _lib.set_tuning(ppw)
    row = []
    for B in (8, 16, 24, 32, 40, 48, 64, 80, 96, 128):
        p, g = synth_inputs(B, 2048, G, 0); pd, gd = torch.from_numpy(p).cuda(), torch.from_numpy(g).cuda()
        pw, pf = d.engine().packed(False)
        film = _lib.film_forward(gd, pf, 3 * L, f, 1e-6, False)
        with torch.no_grad():
            for _ in range(5): _lib.stack_forward(pd, pw, film, 3 * L, f, 0, 1e-6, 'inverse', False)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): _lib.stack_forward(pd, pw, film, 3 * L, f, 0, 1e-6, 'inverse', False)
            e1.record(); torch.cuda.synchronize()
        wgs = B * 2048 // (4 * ppw)
        row.append(f'{wgs}:{e0.elapsed_time(e1) / 30 * 1e3:.0f}')
    print(f'f={f} ppw={ppw}  workgroups:us  ' + '  '.join(row))
_lib.set_tuning(0)
